"""Oracle: result marshalling and bitstring sampling of the reference path.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

Bit-exactness of sampled bitstrings depends on replaying the reference's exact
NumPy call sequence on the *global* ``np.random`` stream,
so these functions deliberately use ``np.random.*`` module-level calls.
"""

from __future__ import annotations

from collections import Counter
from typing import Sequence

import numpy as np

_ONE_STATE = {"ground-rydberg": "r", "digital": "h", "XY": "d"}


def weights(
    state: np.ndarray,
    n_qudits: int,
    eigenbasis: Sequence[str],
    meas_basis: str,
    matching_meas_basis: bool = True,
) -> np.ndarray:
    """``QutipResult._weights`` (pulser_simulation/qutip_result.py:101-158).

    ``state`` is a ket (1-D, length d**N) or a density matrix (d**N x d**N).
    """
    state = np.asarray(state)
    d = len(eigenbasis)
    if state.ndim == 2 and state.shape[0] == state.shape[1] and state.shape[0] > 1:
        probs = np.abs(np.diag(state))  # qutip_result.py:103-104
    else:
        probs = (np.abs(state.reshape(-1, 1)) ** 2).flatten()  # :106
    if d == 2:
        if matching_meas_basis:
            # r-first ordering reversed for ground-rydberg (:114-118)
            w = probs[::-1] if meas_basis == "ground-rydberg" else probs
        else:
            w = np.zeros(probs.size)  # :120-122
            w[0] = 1.0
    elif d in (3, 4):
        if meas_basis not in _ONE_STATE:
            raise RuntimeError(f"Unknown measurement basis '{meas_basis}'.")
        one = list(eigenbasis).index(_ONE_STATE[meas_basis])
        ex_one = [i for i in range(d) if i != one]
        probs = probs.reshape([d] * n_qudits)
        w = np.zeros(2**n_qudits)
        for dec_val in range(2**n_qudits):  # :139-151
            ind = []
            for v in np.binary_repr(dec_val, width=n_qudits):
                ind.append(ex_one if v == "0" else [one])
            w[dec_val] = np.sum(probs[np.ix_(*ind)])
    else:
        raise NotImplementedError(
            "Cannot sample system with single-atom state vectors of "
            "dimension > 4."
        )
    # builtin ``sum`` = sequential left-to-right fp64 accumulation (:158)
    return w / sum(w)


def multinomial(n_samples: int, probs: np.ndarray) -> np.ndarray:
    """``pulser.math.multinomial`` (pulser-core/pulser/math/multinomial.py:18-36)."""
    rnd = np.random.rand(n_samples)
    cum = np.cumsum(probs)
    return np.searchsorted(cum, rnd)


def get_samples(w: np.ndarray, n_samples: int, n_qudits: int) -> Counter:
    """``Result.get_samples`` (pulser-core/pulser/result.py:103-115)."""
    # Insertion order (= order of first occurrence) matters downstream: the
    # SPAM flips iterate over ``list(counter.keys())`` (simresults.py:541).
    return Counter(
        np.binary_repr(i, n_qudits) for i in multinomial(n_samples, w)
    )


def index_from_time(sim_times: np.ndarray, t: float, tol: float = 1e-3) -> int:
    """``SimulationResults._get_index_from_time`` (simresults.py:176-190):
    the FIRST index within ``tol``."""
    try:
        return int(np.where(abs(t - np.asarray(sim_times)) < tol)[0][0])
    except IndexError:
        raise IndexError(
            f"Given time {t} is absent from simulation times within"
            + f" tolerance {tol}."
        )


def spam_flips(sampled: Counter, eps: float, eps_p: float) -> Counter:
    """Measurement-error bit flips of ``CoherentResults.sample_state``
    (pulser_simulation/simresults.py:537-568)."""
    if eps == 0.0 and eps_p == 0:
        return sampled
    shots = list(sampled.keys())
    n_detects = list(sampled.values())
    shot_arr = np.array([list(s) for s in shots], dtype=int)
    flip_probs = np.where(shot_arr == 1, eps_p, eps)
    flip_rep = np.repeat(flip_probs, n_detects, axis=0)
    rnd = np.random.uniform(size=(np.sum(n_detects), len(shot_arr[0])))
    flips = rnd < flip_rep
    new_shots = shot_arr.repeat(n_detects, axis=0) ^ flips
    det = Counter(map(tuple, new_shots))
    return Counter({"".join(map(str, k)): v for k, v in det.items()})


def sample_state(
    states: Sequence[np.ndarray],
    sim_times: np.ndarray,
    t: float,
    n_samples: int,
    n_qudits: int,
    eigenbasis: Sequence[str],
    meas_basis: str,
    matching_meas_basis: bool = True,
    meas_errors: dict | None = None,
) -> Counter:
    """``CoherentResults.sample_state`` (simresults.py:522-568)."""
    idx = index_from_time(sim_times, t)
    w = weights(states[idx], n_qudits, eigenbasis, meas_basis, matching_meas_basis)
    c = get_samples(w, n_samples, n_qudits)
    if meas_errors is None:
        return c
    return spam_flips(c, meas_errors["epsilon"], meas_errors["epsilon_prime"])


def v2_sample(
    state: np.ndarray,
    eigenstates: Sequence[str],
    num_shots: int,
    one_state: str,
    p_false_pos: float = 0.0,
    p_false_neg: float = 0.0,
) -> Counter:
    """``QutipState.sample`` of the V2 backend
    (pulser_simulation/qutip_state.py:112-217): probabilities below
    ``1/(1000*num_shots)`` are dropped and the rest renormalised with
    ``np.sum``; bitstring probabilities are accumulated in basis-state order;
    flips draw ``uniform(size=(num_shots, N))``."""
    state = np.asarray(state)
    d = len(eigenstates)
    if state.ndim == 2 and state.shape[0] == state.shape[1] and state.shape[0] > 1:
        probs = np.abs(np.diag(state)).real
    else:
        probs = (np.abs(state.reshape(-1, 1)) ** 2).flatten().real
    n = int(round(np.log(probs.size) / np.log(d)))
    cutoff = 1 / (1000 * num_shots)
    non_zero = np.argwhere(probs > cutoff).flatten()
    p = probs[non_zero]
    p = p / np.sum(p)
    bit_probs: dict[str, float] = {}
    for idx, pv in zip(non_zero, p):
        digits = np.base_repr(idx, base=d).zfill(n)
        bits = "".join("1" if eigenstates[int(c)] == one_state else "0" for c in digits)
        bit_probs[bits] = bit_probs.get(bits, 0.0) + pv
    bitstrings = np.array(list(bit_probs))
    pr = np.array(list(map(float, bit_probs.values())))
    indices = multinomial(num_shots, pr)
    if p_false_pos == 0.0 and p_false_neg == 0.0:
        return Counter(bitstrings[indices].tolist())
    arr = np.array([list(bs) for bs in bitstrings[indices]], dtype=int)
    flip_probs = np.where(arr == 1, p_false_neg, p_false_pos)
    rnd = np.random.uniform(size=flip_probs.shape)
    new = arr ^ (rnd < flip_probs)
    cnt = Counter(map(tuple, new))
    return Counter({"".join(map(str, k)): v for k, v in cnt.items()})
