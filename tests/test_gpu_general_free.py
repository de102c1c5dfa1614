"""-m gpu: the matrix-free terms of the general path (k_general.hpp: digit decode + gathers for one- and
two-site operators, dense diagonals) against the explicit-CSR terms of the same generator - multi-level
bases, leakage, XY with SLM mask, kets and density matrices, the one-launch kernel and the multi-launch
stepper - and a register beyond the sizes the reference's tests reach."""
from __future__ import annotations

import time

import numpy as np
import pytest

from helpers import load_fixture, with_anneal_samples
from pulser_amd import NoiseModel, QutipEmulator
from pulser_amd import problem as P

pytestmark = pytest.mark.gpu


def _problem_and_state(fixture):
    from pulser_amd.hamiltonian_data import SequenceInputs

    prob, _ = load_fixture(fixture)
    if "inputs" in prob:
        emu = QutipEmulator(SequenceInputs.from_dict(prob["inputs"]), sampling_rate=0.1,
                            noise_model=NoiseModel(dephasing_rate=0.05))
        return emu._current_problem, np.asarray(emu.initial_state).reshape(-1)
    if fixture.startswith("cfg3"):
        prob = with_anneal_samples(prob)
        prob["duration"] = 61  # a slice of the anneal is enough here
        prob["samples"]["Global"]["ground-rydberg"] = {k: np.asarray(v)[400:461] for k, v in
                                                       prob["samples"]["Global"]["ground-rydberg"].items()}
    d, n = len(prob["eigenbasis"]), prob["n_qudits"]
    init = np.zeros(d**n, dtype=complex)
    init[-1 if d == 2 else sum((list(prob["eigenbasis"]).index("g")) * d**k for k in range(n))] = 1.0
    return prob, init


@pytest.mark.parametrize("fixture,mesolve", [("noises_all_0.npz", True), ("noises_all_0.npz", False),
                                             ("noises_all_3.npz", True), ("noisy_xy_0.npz", True),
                                             ("noisy_xy_2.npz", False), ("noises_digital_6.npz", True),
                                             ("noises_digital_2.npz", False), ("cfg3_tri4_dephasing.npz", True)])
@pytest.mark.parametrize("multi", [False, True])
def test_matrix_free_terms_match_csr_terms(fixture, mesolve, multi):
    from pulser_amd.engine import GeneralEngine
    from pulser_amd.general import lower_general

    prob, init = _problem_and_state(fixture)
    T = int(prob["duration"]) - 1
    times = np.array([0.0, 0.3 * T * 1e-3, T * 1e-3])
    outs, gens = [], []
    rng = np.random.default_rng(3)
    for free in (True, False):
        tables = lower_general(prob, mesolve=mesolve, matrix_free=free)
        assert (tables.free is not None) == free
        with GeneralEngine(tables) as eng:
            eng.set_path(multi)
            outs.append(eng.solve(eng.new_state(init), times, tol=1e-13).cpu().numpy())
            if not multi:  # the operator itself: G(t) x on a random vector
                import torch

                x = torch.from_numpy(rng.normal(size=(1, tables.dim)) + 1j * rng.normal(size=(1, tables.dim))).to(eng.device)
                gens.append([eng.apply_generator(x, t).cpu().numpy() for t in (0.0, 0.41 * T * 1e-3)])
        rng = np.random.default_rng(3)
    if gens:
        assert np.max(np.abs(np.asarray(gens[0]) - np.asarray(gens[1]))) < 1e-12 * max(1.0, np.max(np.abs(gens[1])))
    # the two term lists have different norm bounds, hence different steps and Taylor orders: equal within the stepper's tolerance
    assert np.max(np.abs(outs[0] - outs[1])) < 2e-8
    assert np.max(np.abs(outs[0][-1] - outs[0][0])) > 1e-3


def test_matrix_free_three_level_register_of_nine_atoms():
    """3-level 'all' basis (ground-rydberg global + raman local channels) on 9 atoms = 19 683 amplitudes:
    matrix-free terms against CSR terms of the same generator; the host never builds an operator."""
    from pulser_amd.engine import GeneralEngine
    from pulser_amd.general import lower_general

    n, T = 9, 41
    rng = np.random.default_rng(5)
    coords = P.register_coords(P.square_rect(3, 3), 6.5)
    t = np.arange(T) / 1000.0
    prob = P.make_ising_problem(coords, {"amp": 6.0 + 2.0 * np.sin(40 * t), "det": -3.0 + 50 * t, "phase": 0.4 * np.ones(T)})
    prob["eigenbasis"] = ["r", "g", "h"]
    prob["basis_name"] = "all"
    prob["samples"]["Local"] = {"digital": {q: {"amp": rng.uniform(2, 8) * np.ones(T), "det": rng.uniform(-3, 3) * np.ones(T),
                                                 "phase": rng.uniform(0, 1) * np.ones(T)} for q in (0, 4, 7)}}
    init = np.zeros(3**n, dtype=complex)
    init[sum(1 * 3**k for k in range(n))] = 1.0  # |g...g>
    outs, secs = [], []
    for free in (True, False):
        t0 = time.time()
        tables = lower_general(prob, mesolve=False, matrix_free=free)
        lower_s = time.time() - t0
        with GeneralEngine(tables) as eng:
            eng.solve(eng.new_state(init), [0.0, 0.002])  # warm-up
            t0 = time.time()
            outs.append(eng.solve(eng.new_state(init), [0.0, (T - 1) * 1e-3]).cpu().numpy()[-1])
            secs.append((lower_s, time.time() - t0))
    assert np.max(np.abs(outs[0] - outs[1])) < 1e-10
    assert abs(np.linalg.norm(outs[0]) - 1.0) < 1e-9
    assert np.max(np.abs(outs[0][0] - init)) > 1e-2
    print(f"3-level 9 atoms: lowering {secs[0][0]:.2f} s (matrix-free) vs {secs[1][0]:.2f} s (CSR); "
          f"solve {secs[0][1] * 1e3:.0f} ms vs {secs[1][1] * 1e3:.0f} ms")
    # the site-fused application (all terms of a site added into one small matrix per exponential) against the
    # term-by-term kernel: same generator, several times faster (round 2: 34 ms for this solve)
    tables = lower_general(prob, mesolve=False, matrix_free=True)
    res = {}
    xh = rng.normal(size=(1, tables.dim)) + 1j * rng.normal(size=(1, tables.dim))
    for no_sites in (False, True):
        with GeneralEngine(tables) as eng:
            eng.set_path(False, no_sites=no_sites)
            g = eng.apply_generator(eng.torch.from_numpy(xh).to(eng.device), 0.0123).cpu().numpy()
            eng.solve(eng.new_state(init), [0.0, 0.002])
            best = np.inf
            for _ in range(3):  # (the fastest of three: other processes may share the GPU)
                eng.torch.cuda.synchronize()
                t0 = time.time()
                out = eng.solve(eng.new_state(init), [0.0, (T - 1) * 1e-3]).cpu().numpy()[-1]
                best = min(best, time.time() - t0)
            res[no_sites] = (g, out, best)
    assert np.max(np.abs(res[False][0] - res[True][0])) < 1e-12 * np.max(np.abs(res[True][0]))
    assert np.max(np.abs(res[False][1] - res[True][1])) < 1e-12
    print(f"3-level 9 atoms: site-fused {res[False][2] * 1e3:.1f} ms vs term-by-term {res[True][2] * 1e3:.1f} ms")
    import os
    if "PYTEST_XDIST_WORKER" not in os.environ:  # wall-clock ratios mean nothing with several workers on one GPU
        assert res[False][2] < 0.6 * res[True][2]
    else:
        assert res[False][2] < 1.5 * res[True][2]


@pytest.mark.parametrize("case", ["xy12", "all9", "xy8_mesolve", "digital_mesolve"])
def test_padded_site_tables_against_the_round3_kernels(case):
    """k_gen_apply_fused (round 6: padded (value, row offset) site tables, whole vector in LDS where it fits) against
    k_gen_apply_sites (round 3) and the term-by-term kernel: the same generator applied to a random vector at two
    times, and a short solve.  xy12: 66 exchange pairs on 4 096 amplitudes (vector staged in LDS); all9: 3-level
    register, 19 683 amplitudes (gathers from L2); mesolve cases: two-digit superoperator sites on vec(rho)."""
    import torch

    from helpers import three_level_problem, xy_problem
    from pulser_amd.engine import GeneralEngine
    from pulser_amd.general import lower_general

    mesolve = case.endswith("mesolve")
    if case == "xy12":
        prob, init, _ = xy_problem(12)
        t_end = 0.004
    elif case == "all9":
        prob, init, _ = three_level_problem(9, T=41)
        t_end = 0.02
    elif case == "xy8_mesolve":
        prob, init = _problem_and_state("noisy_xy_0.npz")
        t_end = 0.02
    else:
        prob, init = _problem_and_state("noises_digital_6.npz")
        t_end = 0.05
    tables = lower_general(prob, mesolve=mesolve, matrix_free=True)
    rng = np.random.default_rng(11)
    xh = rng.normal(size=(1, tables.dim)) + 1j * rng.normal(size=(1, tables.dim))
    res = {}
    for name, kw in (("fused", {}), ("sites", {"no_fused": True}), ("terms", {"no_sites": True})):
        with GeneralEngine(tables) as eng:
            eng.set_path(True, **kw)
            x = torch.from_numpy(xh).to(eng.device)
            g = [eng.apply_generator(x, t).cpu().numpy() for t in (0.0, 0.37 * t_end)]
            out = eng.solve(eng.new_state(init), [0.0, t_end]).cpu().numpy()[-1]
            res[name] = (np.asarray(g), out, eng.stats())
    scale = max(1.0, float(np.max(np.abs(res["terms"][0]))))
    assert np.max(np.abs(res["fused"][0] - res["terms"][0])) < 1e-12 * scale
    assert np.max(np.abs(res["sites"][0] - res["terms"][0])) < 1e-12 * scale
    assert np.max(np.abs(res["fused"][1] - res["terms"][1])) < 1e-10
    assert np.max(np.abs(res["fused"][1] - res["sites"][1])) < 1e-11
    # (one launch per exponential for coefficients + site matrices instead of two)
    assert res["fused"][2]["n_applications"] == res["sites"][2]["n_applications"]
    if not mesolve:  # something happened on the slice
        assert np.max(np.abs(res["fused"][1] - init)) > 1e-6
