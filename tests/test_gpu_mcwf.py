"""GPU quantum-jump trajectories (``qutip.mcsolve`` on the path,
pulser-simulation/pulser_simulation/simulation.py:705-735).

qutip seeds its own generators, so parity with the reference is statistical;
what CAN be pinned is pinned here:

* the no-jump evolution under H_eff against the tight CPU integration;
* every trajectory (states, number/position of jumps) against the CPU
  restatement of the same algorithm with the same Philox stream
  (``oracle/mcwf.py``);
* the trajectory average against the oracle's ``mesolve`` (itself pinned on the
  reference's golden Counters) within the Monte-Carlo error;
* the front-end: Solver.DEFAULT with stochastic noise and Solver.MCSOLVER.
"""
import numpy as np
import pytest

from helpers import DEPOL_PAULIS, load_fixture, local_problem, with_anneal_samples
from test_host_logic import _inputs_from_problem

from pulser_amd import NoiseModel, QutipEmulator, Solver
from pulser_amd._lib import RydError
from pulser_amd.engine import Engine
from pulser_amd.terms import lower

pytestmark = pytest.mark.gpu

GRID = np.arange(401) / 1000.0
EVAL = np.array([0.0, 0.1, 0.25, 0.4])
OPS = [(np.sqrt(3.0), "sigma_gr"), (np.sqrt(2 * 0.9), "sigma_rr")] + \
      [(np.sqrt(1.2 / 4), p) for p in "xyz"]


def _problem(n, seed=3):
    return local_problem(n, seed=seed, collapse_ops=OPS, paulis=DEPOL_PAULIS)


def test_no_jump_evolution_under_h_eff():
    from oracle import mcwf, qutip_path as qp

    prob = _problem(3)
    ham = qp.build_hamiltonian(prob)
    psi0 = qp.all_ground_state(3, prob["eigenbasis"])
    ref = qp._zvode(mcwf.effective_rhs(ham), psi0, EVAL, qp.TIGHT)
    with Engine(lower([prob]), mode="mcsolve") as eng:
        state = eng.new_state(psi0.reshape(1, -1))
        got = eng.solve(state, EVAL).cpu().numpy()
        assert eng.stats()["n_steps"] == 400  # one CF4 step per sample interval
    for i in range(1, len(EVAL)):
        assert np.max(np.abs(got[i - 1][0] - ref[i])) < 1e-8
    assert np.vdot(ref[-1], ref[-1]).real < 0.7  # the norm really decays


@pytest.mark.parametrize("n,generic", [(2, False), (3, False), (3, True)])
def test_trajectories_match_cpu_restatement(n, generic):
    """Persistent LDS-resident kernel (one launch) and the multi-launch kernels."""
    from oracle import mcwf, qutip_path as qp

    prob = _problem(n, seed=5)
    ham = qp.build_hamiltonian(prob)
    psi0 = qp.all_ground_state(n, prob["eigenbasis"])
    seeds = np.array([1, 2**40 + 17, 123456789012345, 2**64 - 1, 99, 4242], dtype=np.uint64)
    with Engine(lower([prob] * len(seeds)), mode="mcsolve") as eng:
        eng.set_path(generic)
        state = eng.new_state(psi0.reshape(1, -1))
        got = eng.mc_solve(state, EVAL, seeds).cpu().numpy()
        counts = eng.mc_jumps()
        final = state.cpu().numpy()
        assert eng.stats()["n_steps"] == 400
    total = 0
    for b, seed in enumerate(seeds):
        ref, jumps = mcwf.mcwf_trajectory(ham, psi0, GRID, EVAL, int(seed))
        assert counts[b] == len(jumps)
        total += len(jumps)
        for i in range(1, len(EVAL)):
            assert np.max(np.abs(got[i - 1][b] - ref[i])) < 1e-7, (b, i, jumps)
        assert np.max(np.abs(final[b] - ref[-1])) < 1e-7
        assert abs(np.linalg.norm(final[b]) - 1) < 1e-12
    assert total >= len(seeds)  # the test exercises jumps, not only decay


@pytest.mark.parametrize("n", [5, 7, 10, 11, 12, 13])
def test_persistent_and_multi_launch_trajectories_agree(n):
    """Every workgroup shape of the persistent kernel (64 .. 1024 threads, 1-4
    amplitudes per thread) against the tiled multi-launch kernels: same seeds,
    same jumps, same kets."""
    prob = _problem(n, seed=20 + n)
    seeds = np.arange(7, 7 + 12, dtype=np.uint64) * np.uint64(2654435761)
    out = []
    for generic in (False, True):
        with Engine(lower([prob] * len(seeds)), mode="mcsolve") as eng:
            eng.set_path(generic)
            state = eng.new_state()
            snaps = eng.mc_solve(state, np.array([0.0, 0.15, 0.3]), seeds).cpu().numpy()
            out.append((snaps, eng.mc_jumps(), eng.stats()["n_launches"]))
    assert out[0][2] == 1 and out[1][2] > 100
    assert np.array_equal(out[0][1], out[1][1]) and out[0][1].sum() > 0
    assert np.max(np.abs(out[0][0] - out[1][0])) < 1e-10
    assert np.allclose(np.linalg.norm(out[0][0], axis=-1), 1.0, atol=1e-12)


def test_trajectory_is_a_function_of_its_seed_only():
    prob = _problem(4, seed=8)
    seeds = np.arange(100, 116, dtype=np.uint64)
    runs = []
    for sel in (slice(0, 16), slice(5, 9)):
        sd = seeds[sel]
        with Engine(lower([prob] * len(sd)), mode="mcsolve") as eng:
            state = eng.new_state()
            eng.mc_solve(state, EVAL, sd, store=False)
            runs.append((state.cpu().numpy(), eng.mc_jumps()))
    assert np.array_equal(runs[0][0][5:9], runs[1][0])
    assert np.array_equal(runs[0][1][5:9], runs[1][1])


def test_trajectory_average_converges_to_master_equation():
    from oracle import qutip_path as qp

    prob = _problem(3, seed=11)
    ham = qp.build_hamiltonian(prob)
    psi0 = qp.all_ground_state(3, prob["eigenbasis"])
    rho = qp.mesolve(ham, psi0, EVAL, **qp.TIGHT)
    ntraj = 16384
    seeds = np.random.default_rng(0).integers(0, 2**64, size=ntraj, dtype=np.uint64)
    with Engine(lower([prob] * ntraj), mode="mcsolve") as eng:
        state = eng.new_state(psi0.reshape(1, -1))
        snaps = eng.mc_solve(state, EVAL, seeds)
        jumps = eng.mc_jumps()
        acc = eng.torch.zeros((len(EVAL) - 1, 8, 8), dtype=eng.torch.complex128, device=eng.device)
        for i in range(len(EVAL) - 1):
            eng.outer_accumulate(snaps[i], acc[i])
        avg = (acc / ntraj).cpu().numpy()
    assert jumps.mean() > 1.0
    for i in range(1, len(EVAL)):
        r = np.asarray(rho[i]).reshape(8, 8)
        # element-wise Monte-Carlo error: <= 0.5 / sqrt(ntraj) = 0.004; allow 5 sigma
        assert np.max(np.abs(avg[i - 1] - r)) < 0.02, i
        assert abs(np.trace(avg[i - 1]).real - 1) < 1e-12
    assert np.trace(avg[-1] @ avg[-1]).real < 0.9  # a mixed state, not a ket


def test_non_diagonal_decay_is_rejected_by_the_kernels():
    prob = local_problem(2, collapse_ops=[(0.3, np.array([[1.0, 1.0], [0.0, 0.0]]))])
    with pytest.raises(RydError, match="not diagonal"):
        Engine(lower([prob]), mode="mcsolve")


def _tri4_inputs():
    prob, extra = load_fixture("cfg3_tri4_dephasing.npz")
    return _inputs_from_problem(with_anneal_samples(prob), "ground-rydberg")


def _tv(c1, c2):
    keys = set(c1) | set(c2)
    n1, n2 = sum(c1.values()), sum(c2.values())
    return 0.5 * sum(abs(c1.get(k, 0) / n1 - c2.get(k, 0) / n2) for k in keys)


def test_default_solver_with_stochastic_noise_runs_quantum_jumps():
    """simulation.py:705-712: collapse operators + stochastic noise -> mcsolve
    with one jump trajectory per noise trajectory; same distribution as
    integrating the master equation for every noise trajectory."""
    inputs = _tri4_inputs()
    nm = NoiseModel(dephasing_rate=0.3, relaxation_rate=0.2, state_prep_error=0.05,
                    amp_sigma=0.05, samples_per_run=4)
    counts = {}
    for solver in (Solver.DEFAULT, Solver.MESOLVER):
        np.random.seed(21)
        emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=600, solver=solver,
                            evaluation_times="Minimal")
        assert emu._solver_mode(emu._current_problem) == ("mcsolve" if solver == Solver.DEFAULT
                                                          else "mesolve")
        with pytest.warns(DeprecationWarning):
            res = emu.run(seeds=5)
        assert res.n_measures == 2400
        counts[solver] = res[-1].bitstring_counts
        if solver == Solver.DEFAULT:
            assert emu.last_mc_jumps.sum() > 0
            first = dict(counts[solver])
            np.random.seed(21)
            emu2 = QutipEmulator(inputs, noise_model=nm, n_trajectories=600,
                                 evaluation_times="Minimal")
            with pytest.warns(DeprecationWarning):
                assert dict(emu2.run(seeds=5)[-1].bitstring_counts) == first  # reproducible
    # 16 outcomes, 2400 samples each: the total-variation distance of two draws
    # of one distribution is ~ 0.03-0.05
    assert _tv(counts[Solver.DEFAULT], counts[Solver.MESOLVER]) < 0.1


def test_mcsolver_deterministic_run_returns_the_averaged_density_matrix():
    """simulation.py:843: ntraj = n_trajectories; result.states are the
    trajectory-averaged density matrices."""
    inputs = _tri4_inputs()
    nm = NoiseModel(dephasing_rate=0.3, relaxation_rate=0.2)
    ref = QutipEmulator(inputs, noise_model=nm, evaluation_times="Minimal")
    with pytest.warns(DeprecationWarning):
        exact = ref.run()
    rho = np.asarray(exact.states[-1])
    emu = QutipEmulator(inputs, noise_model=nm, solver=Solver.MCSOLVER, n_trajectories=3000,
                        evaluation_times="Minimal")
    with pytest.warns(DeprecationWarning):
        res = emu.run(seeds=1)
    got = np.asarray(res.states[-1])
    assert got.shape == (16, 16) and len(emu.last_mc_jumps) == 3000
    assert abs(np.trace(got).real - 1) < 1e-12 and np.allclose(got, got.conj().T)
    assert np.max(np.abs(got - rho)) < 5 * 0.5 / np.sqrt(3000)
    assert np.allclose(np.asarray(res.states[0]), np.asarray(exact.states[0]))
    assert sum(res.sample_final_state(500).values()) == 500


def test_exotic_collapse_operators_fall_back_to_the_master_equation():
    """sum C^dag C not diagonal: no quantum jumps on the ket kernels - the master
    equation (the average mcsolve estimates) is integrated instead."""
    inputs = _tri4_inputs()
    op = np.array([[1.0, 1.0], [0.0, 0.0]], dtype=complex)
    nm = NoiseModel(eff_noise_opers=[op], eff_noise_rates=[0.2])
    a = QutipEmulator(inputs, noise_model=nm, evaluation_times="Minimal")
    b = QutipEmulator(inputs, noise_model=nm, solver=Solver.MCSOLVER, n_trajectories=5,
                      evaluation_times="Minimal")
    assert not b._mc_fast_ok(b._current_problem)
    with pytest.warns(DeprecationWarning):
        ra, rb = a.run(), b.run()
    assert np.allclose(np.asarray(ra.states[-1]), np.asarray(rb.states[-1]), atol=1e-12)


def test_quantum_jumps_on_register_tile_and_tiled_kernels_agree():
    """14 atoms (beyond the persistent kernel): the 2^14 register-tile kernel, the
    tiled kernel's single-launch plan and its two-pass plan, all with the H_eff decay
    diagonal, give the same trajectories; norms stay 1 and jumps happen."""
    prob = _problem(14, seed=31)
    seeds = np.arange(64, dtype=np.uint64) + np.uint64(5)
    tables = lower([prob] * 64)
    out = []
    for no14, no_single in ((False, False), (True, False), (True, True)):
        with Engine(tables, mode="mcsolve") as eng:
            eng.set_path(False, no_tile14=no14, no_single_pass=no_single)
            state = eng.new_state()
            # (the default for quantum jumps at 14+ atoms is the split-operator path, tested below)
            eng.mc_solve(state, np.array([0.0, 0.03]), seeds, store=False, method="taylor")
            out.append((state.cpu().numpy(), eng.mc_jumps(), eng.stats()["passes"]))
    assert [o[2] for o in out] == [1, 1, 2]
    for other in out[1:]:
        assert np.array_equal(out[0][1], other[1]) and out[0][1].sum() > 10
        assert np.max(np.abs(out[0][0] - other[0])) < 1e-10
    assert np.allclose(np.linalg.norm(out[0][0], axis=-1), 1.0, atol=1e-12)


@pytest.mark.parametrize("n,batch", [(14, 16), (16, 6), (20, 2)])
def test_quantum_jumps_on_the_split_operator_passes_match_the_taylor_path(n, batch):
    """Collapse operators on the split-operator ket passes (the default from 14 atoms on): H_eff's decay
    diagonal is a real factor of the D stages, the jump bookkeeping runs between schedule steps exactly as
    on the CF4 + Taylor path - same seeds, same jumps, kets within the parity bar."""
    strong = [(np.sqrt(6.0), "sigma_gr"), (np.sqrt(2 * 1.5), "sigma_rr")] + [(np.sqrt(2.0 / 4), p) for p in "xyz"]
    prob = local_problem(n, seed=40 + n, duration=61, collapse_ops=strong, paulis=DEPOL_PAULIS)
    seeds = np.arange(batch, dtype=np.uint64) * np.uint64(7919) + np.uint64(11)
    times = np.array([0.0, 0.02, 0.06])
    out = {}
    for method in ("auto", "taylor"):
        with Engine(lower([prob] * batch), mode="mcsolve") as eng:
            state = eng.new_state()
            snaps = eng.mc_solve(state, times, seeds, method=method, tol=0.0 if method == "auto" else 1e-12).cpu().numpy()
            out[method] = (snaps, eng.mc_jumps(), eng.stats())
    assert out["auto"][2]["passes"] in (1, 2) and out["auto"][2]["n_applications"] < out["taylor"][2]["n_applications"]
    assert np.array_equal(out["auto"][1], out["taylor"][1]) and out["auto"][1].sum() >= batch // 2
    assert np.max(np.abs(out["auto"][0] - out["taylor"][0])) < 1e-7
    assert np.allclose(np.linalg.norm(out["auto"][0], axis=-1), 1.0, atol=1e-12)


def test_split_operator_no_jump_decay_against_the_tight_oracle():
    """No-jump evolution under H_eff on the split-operator passes (forced at 8 atoms, where zvode is cheap)."""
    from oracle import mcwf, qutip_path as qp

    n = 8
    prob = local_problem(n, seed=3, duration=61, collapse_ops=OPS, paulis=DEPOL_PAULIS)
    ham = qp.build_hamiltonian(prob)
    psi0 = qp.all_ground_state(n, prob["eigenbasis"])
    times = np.array([0.0, 0.03, 0.06])
    ref = qp._zvode(mcwf.effective_rhs(ham), psi0, times, qp.TIGHT)
    with Engine(lower([prob]), mode="mcsolve") as eng:
        state = eng.new_state()
        got = eng.solve(state, times, method="split").cpu().numpy()
    for i in (1, 2):
        assert np.max(np.abs(got[i - 1][0] - ref[i])) < 1e-8
    assert np.vdot(ref[-1], ref[-1]).real < 0.95
