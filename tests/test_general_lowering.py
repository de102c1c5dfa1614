"""CPU: the matrix-free lowering of the general path (one- / two-site operators with digit strides,
dense diagonals; pulser_amd/general.py) against the explicit-CSR lowering of the same problem -
the generator G(t) as a dense matrix at several times, for every multi-level / XY / leakage fixture
and the 2-level ones, kets and density matrices."""
from __future__ import annotations

import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN as GOLDEN_DIR, load_fixture
from pulser_amd.general import dense_generator, lower_general


def _problems():
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(path)
        try:
            prob, _ = load_fixture(name)
        except Exception:
            continue
        if not isinstance(prob, dict):
            continue
        if "eigenbasis" in prob and len(prob["eigenbasis"]) ** int(prob["n_qudits"]) > 4096:
            continue  # dense generators only (the 14-atom headline fixture is not for this test)
        if "inputs" in prob:  # XY fixtures store the sequence inputs
            yield name
        elif "eigenbasis" in prob:
            yield name


def _problem(name):
    prob, _ = load_fixture(name)
    if "inputs" in prob:
        from pulser_amd import NoiseModel, QutipEmulator
        from pulser_amd.hamiltonian_data import SequenceInputs

        emu = QutipEmulator(SequenceInputs.from_dict(prob["inputs"]), sampling_rate=0.1,
                            noise_model=NoiseModel(dephasing_rate=0.05))
        return emu._current_problem
    return prob


def _coefs(tables, t):
    """coef_t(t) of every term from the tables' own spline pieces."""
    idx = int(np.clip(np.searchsorted(tables.tknots, t, side="right") - 1, 0, len(tables.tknots) - 2))
    u = t - tables.tknots[idx]
    out = []
    for i in range(len(tables.values)):
        c = 1.0 + 0j
        if tables.series[i] >= 0:
            p = tables.pp[tables.series[i], idx]
            c = ((p[0] * u + p[1]) * u + p[2]) * u + p[3]
            if tables.conj[i]:
                c = np.conj(c)
        out.append(c * tables.scale[i])
    return out


NAMES = list(_problems())


def test_fixture_list_covers_multilevel_and_xy():
    assert len(NAMES) >= 30
    assert any("xy" in n for n in NAMES) and any("digital" in n for n in NAMES) and any("all" in n for n in NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_matrix_free_terms_equal_the_csr_generator(name):
    prob = _problem(name)
    d, n = len(prob["eigenbasis"]), int(prob["n_qudits"])
    done = 0
    for mesolve in (False, True):
        if d ** (2 * n if mesolve else n) > 4096:
            continue
        free = lower_general(prob, mesolve, matrix_free=True)
        csr = lower_general(prob, mesolve, matrix_free=False)
        assert free.free is not None and all(f is not None for f in free.free)
        assert csr.free is None
        T = (int(prob["duration"]) - 1) * 1e-3
        for t in ((0.0, 0.37 * T, T) if free.dim <= 1024 else (0.37 * T,)):
            a = dense_generator(free, _coefs(free, t))
            b = dense_generator(csr, _coefs(csr, t))
            assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(b)))
        # the norm bounds the step planner uses must dominate the true row sums: term by term for the time-dependent
        # terms, JOINTLY for the time-independent ones (round 6: their norms are scaled so that they add up to the exact
        # largest row sum of sum_t |A_t| instead of the sum of the separate maxima - general._tighten_static_norms)
        static = [i for i in range(len(free.free)) if free.series[i] < 0]
        for i, f in enumerate(free.free if free.dim <= 1024 else []):
            if i in static and len(static) > 1:
                continue
            one = [0.0] * len(free.free)
            one[i] = 1.0
            assert np.abs(dense_generator(free, one)).sum(axis=1).max() <= free.row_norm[i] * (1 + 1e-12) + 1e-300
        if free.dim <= 1024 and len(static) > 1:
            rows = sum(np.abs(dense_generator(free, [1.0 if j == i else 0.0 for j in range(len(free.free))])) for i in static)
            joint = float(sum(free.row_norm[i] for i in static))
            exact = float(rows.sum(axis=1).max())
            assert exact <= joint * (1 + 1e-12) + 1e-300
            assert joint <= exact * (1 + 1e-9) + 1e-300  # ... and is that row sum, not more
        done += 1
    assert done >= 1


def test_matrix_free_lowering_has_no_size_limit_of_the_operator():
    """3-level register of 9 atoms (19 683 amplitudes): the CSR lowering would build kron products; the
    matrix-free one only produces (d x d) matrices, strides and one diagonal."""
    prob, _ = load_fixture("noises_all_0.npz")
    free = lower_general(prob, False, matrix_free=True)
    assert lower_general(prob, False).free is None  # small systems default to explicit CSR terms
    sizes = sum(f[1].nbytes if f[0] == "diag" else f[7].nbytes + f[3].nbytes for f in free.free)
    assert sizes < 64 * free.dim  # a few vectors at most
