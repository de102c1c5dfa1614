"""Shared helpers for the parity tests (oracle = checker only)."""
from __future__ import annotations

import os

import numpy as np

from pulser_amd import problem as P

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    prob, extra = P.load_problem(os.path.join(GOLDEN, name))
    return prob, extra


def with_anneal_samples(prob):
    """cfg2/cfg3 fixtures store no samples: regenerate them synthetically."""
    s = P.anneal_samples()
    prob = dict(prob)
    prob["samples"] = {"Global": {"ground-rydberg": s}, "Local": {}}
    return prob


def blockade_radius():
    return (P.C6_LEVEL70 / (4 * 2 * np.pi / 2)) ** (1 / 6)


def chain_problem(n, collapse_ops=None):
    coords = P.register_coords(P.square_rect(1, n), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=collapse_ops)


def local_problem(n, seed=0, duration=401, collapse_ops=None, paulis=None):
    """Random smooth per-qubit drives/detunings/phases (Local addressing)."""
    rng = np.random.default_rng(seed)
    t = np.arange(duration) / 1000.0
    coords = P.register_coords(P.square_rect(1, n), 7.0) + rng.normal(0, 0.3, (n, 2))
    prob = P.make_ising_problem(coords, {"amp": np.zeros(duration), "det": np.zeros(duration), "phase": np.zeros(duration)})
    loc = {}
    for q in range(n):
        a, b, c = rng.uniform(2, 12, 3)
        loc[q] = {
            "amp": a * (1 + 0.5 * np.sin(2 * np.pi * (q + 1) * t / t[-1])),
            "det": b * np.cos(3 * t + q) - c,
            "phase": 0.3 * q + 0.8 * np.sin(5 * t),
        }
    prob["samples"] = {"Global": {}, "Local": {"ground-rydberg": loc}}
    prob["collapse_ops"] = list(collapse_ops or [])
    prob["depolarizing_pauli_2ds"] = dict(paulis or {})
    return prob


DEPOL_PAULIS = {
    "x": [(1, "sigma_gr"), (1, "sigma_rg")],
    "y": [(1j, "sigma_gr"), (-1j, "sigma_rg")],
    "z": [(1, "sigma_rr"), (-1, "sigma_gg")],
}


def rand_state(dim, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=dim) + 1j * rng.normal(size=dim)
    return v / np.linalg.norm(v)
