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


def tight_density_matrices(extra, n):
    """cfg3 fixtures written by ``gen_cfg3_tight`` keep the upper triangle only."""
    D = 2**n
    iu = np.triu_indices(D)
    out = []
    for vals in np.asarray(extra["oracle_states_tight_triu"]):
        rho = np.zeros((D, D), complex)
        rho[iu] = vals
        low = rho.conj().T.copy()
        low[np.diag_indices(D)] = 0.0
        out.append(rho + low)
    return np.stack(out)


def sketch_rows(D, rows=32, seed=11):
    """The rows / probe vectors of ``make_fixtures.sketch_density_matrices`` (same seeded draws)."""
    rng = np.random.default_rng(seed)
    return np.unique(np.concatenate([[0, D - 1], rng.choice(D, min(D, rows) - 2, replace=False)]))


def sketch_probes(D, probes=4, seed=11):
    rng = np.random.default_rng(seed + 1)
    return rng.standard_normal((D, probes)) + 1j * rng.standard_normal((D, probes))


def sketch_errors(rho, extra, k):
    """Max-abs deviation of ``rho`` from the tight oracle at stored time ``k`` over everything the fixture
    keeps: 32 full rows, the diagonal, rho @ (4 Gaussian probes) - which sees an error anywhere - and purity."""
    D = rho.shape[0]
    sk = extra["sketch"]
    rows = sketch_rows(D, sk["rows"], sk["seed"])
    if sk.get("keep"):  # (12 atoms: every 4th of the 32 drawn rows - a 2^12-entry row is 64 KiB per stored time)
        rows = rows[np.round(np.linspace(0, len(rows) - 1, int(sk["keep"]))).astype(int)]
    assert np.array_equal(rows, np.asarray(extra["oracle_rows"]))
    V = sketch_probes(D, sk["probes"], sk["seed"])
    return {
        "rows": float(np.max(np.abs(rho[rows] - np.asarray(extra["oracle_rows_tight"])[k]))),
        "diag": float(np.max(np.abs(np.diag(rho) - np.asarray(extra["oracle_diag_tight"])[k]))),
        "probes": float(np.max(np.abs(rho @ V - np.asarray(extra["oracle_probe_products_tight"])[k]))),
        "purity": float(abs(np.vdot(rho, rho).real - np.asarray(extra["oracle_purity_tight"])[k])),
    }
