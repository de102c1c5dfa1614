"""GPU end-to-end: the QutipEmulator-compatible front-end on the HIP engine
against the reference's seeded golden values and the oracle's Counters.

Bit-exact bitstring Counters (sampling indices) for given seeds; states within
1e-7 of the tight oracle; the reference's own state golden at its rtol 1e-2.
"""
from collections import Counter

import numpy as np
import pytest

from helpers import load_fixture, with_anneal_samples
from test_host_logic import _all_basis_emulator, _chain12_inputs, _inputs_from_problem

from pulser_amd import NoiseModel, QutipEmulator, Solver
from pulser_amd import problem as P
from pulser_amd.hamiltonian_data import single_global_channel

pytestmark = pytest.mark.gpu

SUPPORTED_RYDBERG = list(range(7))  # index 6 = leakage (3-level): explicit-term general path


@pytest.mark.parametrize("k", SUPPORTED_RYDBERG)
def test_reference_golden_counters_rydberg_end_to_end(k):
    """tests/pulser_simulation/test_simulation.py:978-1040 with the real solver."""
    prob, extra = load_fixture(f"noises_rydberg_{k}.npz")
    noise = tuple(extra["noise"])
    params = {}
    if "dephasing" in noise:
        params.update(dephasing_rate=0.05, hyperfine_dephasing_rate=1e-3)
    if "relaxation" in noise:
        params.update(relaxation_rate=0.01)
    if "depolarizing" in noise:
        params.update(depolarizing_rate=0.05)
    leak = "leakage" in noise
    if leak or "eff_noise" in noise:
        params.update(eff_noise_opers=[np.diag([1.0, 0, 0]).astype(complex) if leak
                                       else np.diag([1.0, -1.0]).astype(complex)],
                      eff_noise_rates=[0.1 if leak else 0.025])
    np.random.seed(123)
    emu = QutipEmulator(_inputs_from_problem(prob, "ground-rydberg"), sampling_rate=0.01,
                        noise_model=NoiseModel(with_leakage=leak, **params))
    with pytest.warns(DeprecationWarning):
        res = emu.run()
    assert res.sample_final_state() == Counter(extra["reference_golden_counter"])
    final = np.asarray(res.states[-1])
    assert np.max(np.abs(final - extra["oracle_final_state_tight"])) < 1e-7
    tr2 = np.trace(final @ final).real
    assert tr2 < 1 and not np.isclose(tr2, 1)


def test_cfg1_plumbing_and_three_atom_golden_state():
    prob, extra = load_fixture("cfg1_square4_pi.npz")
    coords = P.register_coords(P.square_rect(2, 2), 5.0)
    amp = P.blackman_samples(1000, np.pi)
    inputs = single_global_channel(coords, {"amp": amp, "det": 0 * amp, "phase": 0 * amp},
                                   P.C6_LEVEL70, extended=False)
    emu = QutipEmulator(inputs)
    assert len(emu.evaluation_times) == 1001
    with pytest.warns(DeprecationWarning):
        res = emu.run()
    idx = np.asarray(extra["oracle_state_indices"])
    ref = np.asarray(extra["oracle_states_tight"])
    for j, i in enumerate(idx):
        assert np.max(np.abs(np.asarray(res.states[i])[:, 0] - ref[j])) < 1e-7
    np.random.seed(123)
    assert res.sample_final_state(1000) == Counter(extra["oracle_counter_default"])
    h0 = np.asarray(emu.get_hamiltonian(500))
    assert np.allclose(h0, h0.conj().T) and abs(h0[0, 0] - prob["interaction_matrix"][0][np.triu_indices(4, 1)].sum()) < 1e-9

    # test_simulation.py:2156-2190
    prob, extra = load_fixture("three_atom_state.npz")
    emu = QutipEmulator(_inputs_from_problem(prob, "ground-rydberg", c6=P.C6_LEVEL60))
    emu.set_initial_state(np.ones(8))
    np.random.seed(123)
    with pytest.warns(DeprecationWarning):
        res = emu.run()
    final = np.asarray(res.get_final_state())[:, 0]
    assert np.all(np.isclose(final, extra["reference_golden_state"], 1e-2))
    assert res._get_index_from_time(4.0) == 3999


def test_spam_state_prep_trajectories_mesolve_counts():
    """SPAM state-prep errors + dephasing with Solver.MESOLVER: deduplicated
    trajectories, batched on the GPU, sampled in the reference's serial order."""
    prob, extra = load_fixture("cfg3_tri4_dephasing.npz")
    prob = with_anneal_samples(prob)
    inputs = _inputs_from_problem(prob, "ground-rydberg")
    nm = NoiseModel(dephasing_rate=0.05, state_prep_error=0.1, p_false_pos=0.01,
                    p_false_neg=0.05, samples_per_run=5)
    np.random.seed(11)
    emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=12, solver=Solver.MESOLVER,
                        evaluation_times="Minimal")
    reps = [p["reps"] for p in emu._problems]
    assert sum(reps) == 12 and len(reps) > 1
    with pytest.warns(DeprecationWarning):
        res = emu.run()
    assert res.n_measures == 60
    assert sum(res[-1].bitstring_counts.values()) == 60
    # same seed, same counts (deterministic RNG order), and a fresh draw on re-run
    np.random.seed(11)
    emu2 = QutipEmulator(inputs, noise_model=nm, n_trajectories=12, solver=Solver.MESOLVER,
                         evaluation_times="Minimal")
    with pytest.warns(DeprecationWarning):
        res2 = emu2.run()
    assert res2[-1].bitstring_counts == res[-1].bitstring_counts


def test_noisy_doppler_amplitude_trajectories_against_oracle():
    """cfg4 physics, 3 trajectories: every trajectory state equals the oracle's
    tight solution of the noisy problem captured from pulser-core."""
    prob0, extra = load_fixture("cfg4_chain12_noise.npz")
    nm = NoiseModel(**extra["noise_model"])
    np.random.seed(0)
    emu = QutipEmulator(_chain12_inputs(extra), noise_model=nm, n_trajectories=1024,
                        evaluation_times=list(extra["eval_times"]))
    assert len(emu._problems) == 1024
    emu._problems = emu._problems[:3]
    ref = np.asarray(extra["oracle_states_tight"])
    got = emu._solve_batch(emu._problems, False, {})
    for b in range(3):
        for i in range(len(extra["eval_times"])):
            assert np.max(np.abs(np.asarray(got[b].states[i])[:, 0] - ref[b][i])) < 1e-7
    # round 4: a 12-atom batch whose schedule is mostly multi-knot steps takes the register-resident split-operator
    # kernel (a launch per closed run and its coefficient table) instead of the one persistent launch of k_traj
    assert 1 <= emu.last_engine_stats["n_launches"] < 100


def test_cfg4_noisy_run_factored_equals_general_path():
    """Full stochastic run (doppler + amplitude + SPAM, 24 trajectories): the
    factored lowering (shared spline tables + per-atom scales) and the general
    lowering of every noisy problem give the same sampled Counters."""
    _, extra = load_fixture("cfg4_chain12_noise.npz")
    nm = NoiseModel(samples_per_run=10, **extra["noise_model"])
    counts = []
    for general in (False, True):
        np.random.seed(5)
        emu = QutipEmulator(_chain12_inputs(extra), noise_model=nm, n_trajectories=24,
                            evaluation_times="Minimal")
        if general:
            emu._hamiltonian_data.factorable = lambda: False
        with pytest.warns(DeprecationWarning):
            res = emu.run()
        assert res.n_measures == 240
        counts.append([dict(r.bitstring_counts) for r in res])
    assert counts[0] == counts[1]
    assert sum(counts[0][-1].values()) == 240


@pytest.mark.parametrize("k", range(7))
def test_reference_golden_counters_digital_end_to_end(k):
    """tests/pulser_simulation/test_simulation.py:1079-1160 with the real solver:
    digital basis, local Raman pulses, 2-level (tuned kernels) and 3-level with
    leakage / general eff_noise (explicit-term general path)."""
    prob, extra = load_fixture(f"noises_digital_{k}.npz")
    noise = tuple(extra["noise"])
    params = {}
    if "dephasing" in noise:
        params.update(dephasing_rate=0.05, hyperfine_dephasing_rate=0.05)
    if "depolarizing" in noise:
        params.update(depolarizing_rate=0.05)
    leak = "leakage" in noise
    if leak or "eff_noise" in noise:
        params["eff_noise_opers"] = [np.diag([0, 1.0, 0]).astype(complex) if leak
                                     else np.diag([1.0, -1.0]).astype(complex)]
        params["eff_noise_rates"] = [0.1 if leak else 0.025]
    np.random.seed(123)
    emu = QutipEmulator(_inputs_from_problem(prob, "digital"), sampling_rate=0.01,
                        noise_model=NoiseModel(with_leakage=leak, **params))
    assert emu._fast_path_ok(emu._current_problem) == (not leak)
    with pytest.warns(DeprecationWarning):
        res = emu.run()
    assert res.sample_final_state() == Counter(extra["reference_golden_counter"])
    final = np.asarray(res.states[-1])
    assert np.max(np.abs(final - extra["oracle_final_state_tight"])) < 1e-7
    if leak:  # test_simulation.py:1036-1039: nothing leaks into the error state
        st = np.asarray(res.get_final_state())
        assert st.shape == (27, 27)


def test_general_path_matches_tuned_kernels_and_oracle_generator():
    """The explicit-term path against the matrix-free kernels (2-level problem
    forced through it) and against the oracle's Lindblad RHS for a non-diagonal
    eff_noise operator that only the general path supports."""
    from helpers import local_problem
    from oracle import qutip_path as qp
    from pulser_amd.engine import Engine, GeneralEngine
    from pulser_amd.general import lower_general
    import torch

    prob = local_problem(4, seed=2, duration=41)
    times = np.array([0.0, 0.013, 0.04])
    with Engine.from_problems([prob]) as eng:
        st = eng.new_state()
        ref = eng.solve(st, times).cpu().numpy()[:, 0]
    with GeneralEngine(lower_general(prob, mesolve=False)) as g:
        psi0 = np.zeros(16, complex); psi0[-1] = 1
        got = g.solve(g.new_state(psi0), times).cpu().numpy()[:, 0]
    assert np.max(np.abs(got - ref)) < 1e-8  # different norm bounds -> different step/order choices
    # general eff_noise: C = sqrt(rate) * (X + 0.3 Z) has pair-dependent single flips
    op = np.array([[0.3, 1.0], [1.0, -0.3]], dtype=complex)
    prob = local_problem(3, seed=5, duration=41, collapse_ops=[(np.sqrt(0.2), op)])
    assert not QutipEmulator._fast_path_ok(prob)
    ham = qp.build_hamiltonian(prob)
    rhs = qp.lindblad_rhs(ham)
    rng = np.random.default_rng(1)
    x = rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8))
    with GeneralEngine(lower_general(prob, mesolve=True)) as g:
        dev = torch.from_numpy(x.reshape(1, -1).copy()).to(g.device)
        out = g.apply_generator(dev, 0.0123).cpu().numpy()[0].reshape(8, 8)
        assert np.max(np.abs(out - rhs(0.0123, x.ravel()).reshape(8, 8))) < 1e-11
        psi0 = np.zeros(8, complex); psi0[-1] = 1
        rho = g.solve(g.new_state(psi0), np.array([0.0, 0.04])).cpu().numpy()[-1, 0].reshape(8, 8)
    ref = qp.mesolve(ham, psi0, np.array([0.0, 0.04]), max_step=1e-3, **qp.TIGHT)[-1]
    assert np.max(np.abs(rho - ref)) < 1e-7


@pytest.mark.parametrize("k", range(6))
def test_reference_golden_counters_all_basis_end_to_end(k):
    """test_simulation.py:1179-1300 with the real solver: 3-level "all" basis
    (and 4-level with leakage), local + global channels, dephasing / relaxation /
    eff_noise - explicit-term general path."""
    emu, prob, extra = _all_basis_emulator(k)
    assert not emu._fast_path_ok(emu._current_problem)
    np.random.seed(123)
    with pytest.warns(DeprecationWarning):
        res = emu.run()
    assert res.sample_final_state() == Counter(extra["reference_golden_counter"])
    final = np.asarray(res.states[-1])
    assert np.max(np.abs(final - extra["oracle_final_state_tight"])) < 1e-7
    tr2 = np.trace(final @ final).real
    assert tr2 < 1 and not np.isclose(tr2, 1)


@pytest.mark.parametrize("k", range(6))
def test_reference_golden_counters_xy_end_to_end(k):
    """test_simulation.py:1536-1690 (MESOLVER cases) with the real solver: XY
    exchange interaction, SLM-mask switching terms, SPAM trajectories, 2- and
    3-level (leakage) - explicit-term general path."""
    from test_host_logic import _xy_emulator

    emu, extra = _xy_emulator(k)
    with pytest.warns(DeprecationWarning):
        r = emu.run()
    assert r.sample_final_state() == Counter(extra["reference_golden_counter"])


def test_get_hamiltonian_reference_goldens():
    """tests/pulser_simulation/test_simulation.py:477-600: H(t) entries for the
    noiseless case, doppler noise (seed 123) and doppler + register noise (seed
    456) - pins the RNG draw order, the sign conventions, the noisy interaction
    matrix and the spline evaluation through the device generator kernel."""
    coords = np.array([[10.0, 0.0], [0.0, 0.0]])
    coords = coords - coords.mean(axis=0)  # Register.from_coordinates(center=True)
    amp = P.ramp_samples(1500, 0.0, 2.0)
    inputs = single_global_channel(coords, {"amp": amp, "det": np.full(1500, 1.0), "phase": 0 * amp},
                                   P.C6_LEVEL70, extended=False, prefix="atom")
    sim = QutipEmulator(inputs, sampling_rate=0.01)
    with pytest.raises(ValueError, match="less than or equal to"):
        sim.get_hamiltonian(1650)
    with pytest.raises(ValueError, match="greater than or equal to"):
        sim.get_hamiltonian(-10)
    ham = np.asarray(sim.get_hamiltonian(143))
    assert np.isclose(ham[0, 0], P.C6_LEVEL70 / 10**6 - 2 * 1.0)

    np.random.seed(123)
    noisy = QutipEmulator(inputs, noise_model=NoiseModel(samples_per_run=1, temperature=20000),
                          n_trajectories=15)
    np.testing.assert_allclose(
        np.asarray(noisy.get_hamiltonian(144)),
        np.array([[4.47984523, 0.09606404, 0.09606404, 0.0],
                  [0.09606404, 12.03082372, 0.0, 0.09606404],
                  [0.09606404, 0.0, -12.97113702, 0.09606404],
                  [0.0, 0.09606404, 0.09606404, 0.0]], dtype=complex), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(np.asarray(noisy.get_hamiltonian(144, noiseless=True)),
                               np.asarray(QutipEmulator(inputs).get_hamiltonian(144)), atol=1e-12)

    np.random.seed(456)
    noisy = QutipEmulator(
        inputs, noise_model=NoiseModel(samples_per_run=1, temperature=50.0, trap_depth=150.0,
                                       trap_waist=1.0), n_trajectories=1)
    np.testing.assert_allclose(
        np.asarray(noisy.get_hamiltonian(144)),
        np.array([[4.92294305, 0.09606404, 0.09606404, 0.0],
                  [0.09606404, -0.59902269, 0.0, 0.09606404],
                  [0.09606404, 0.0, -0.70099956, 0.09606404],
                  [0.0, 0.09606404, 0.09606404, 0.0]], dtype=complex), rtol=1e-7, atol=1e-9)


def test_dmm_channel_trajectories_against_oracle():
    """DMM + global Rydberg channel (4 atoms), noiseless and with dmm_sigma +
    crosstalk + doppler + SPAM trajectories captured from pulser-core: the GPU
    states equal the oracle's tight solutions of pulser-core's noisy samples."""
    from pulser_amd.hamiltonian_data import SequenceInputs

    prob, extra = load_fixture("dmm_square4.npz")
    inputs = SequenceInputs.from_dict(prob["inputs"])
    times = list(extra["eval_times"])
    np.random.seed(77)
    emu = QutipEmulator(inputs, evaluation_times=times)
    with pytest.warns(DeprecationWarning):
        res = emu.run()
    for i in range(len(times)):
        assert np.max(np.abs(np.asarray(res.states[i])[:, 0] - extra["noiseless_states"][0][i])) < 1e-7
    np.random.seed(77)
    emu = QutipEmulator(inputs, noise_model=NoiseModel(**extra["noisy_model"]), evaluation_times=times)
    assert len(emu._problems) == 5
    got = emu._solve_batch(emu._problems, False, {})
    for b in range(5):
        for i in range(len(times)):
            assert np.max(np.abs(np.asarray(got[b].states[i])[:, 0] - extra["noisy_states"][b][i])) < 1e-7


@pytest.mark.filterwarnings("ignore::DeprecationWarning")
def test_set_config_changes_the_hamiltonian():
    """tests/pulser_simulation/test_simulation.py:828-866: doppler noise moves the
    detuning entries, amplitude noise the drive entries; a 3-level (leakage)
    configuration switches get_hamiltonian to the explicit-term engine."""
    from pulser_amd import SimConfig
    from test_host_logic import _two_atom_inputs

    np.random.seed(123)
    sim = QutipEmulator(_two_atom_inputs(), config=SimConfig(noise="SPAM"))
    sim.reset_config()
    clean = np.asarray(sim.get_hamiltonian(123))
    sim.set_config(SimConfig(noise="doppler", temperature=10000))
    noisy = np.asarray(sim.get_hamiltonian(123))
    assert noisy[0, 0] != clean[0, 0] and noisy[3, 3] == clean[3, 3]
    sim.set_config(SimConfig(noise="amplitude"))
    noisy_amp = np.asarray(sim.get_hamiltonian(123))
    assert noisy_amp[0, 0] == clean[0, 0] and noisy_amp[0, 1] != clean[0, 1]
    z3 = np.diag([1.0, -1.0, 0.0]).astype(complex)
    sim.set_config(SimConfig(noise=("leakage", "eff_noise"), eff_noise_opers=[z3], eff_noise_rates=[0.1]))
    h3 = np.asarray(sim.get_hamiltonian(123))
    assert h3.shape == (9, 9) and np.allclose(h3, h3.conj().T)
    # (r, g, x) x (r, g, x): the 2-level block is the clean Hamiltonian
    keep = [0, 1, 3, 4]
    assert np.allclose(h3[np.ix_(keep, keep)], clean, atol=1e-12)


def test_reference_golden_counter_test_noise_end_to_end(capsys):
    """test_simulation.py:889-923 with the real solver on the GPU (3-level
    "all" basis, 15 SPAM trajectories, seed 3)."""
    from test_host_logic import _spam_all_emulator

    emu, extra = _spam_all_emulator()
    with pytest.warns(DeprecationWarning):
        r = emu.run(print_progress=True)
    assert capsys.readouterr().out.rstrip("\n").split("\n") == [
        "Emulating Trajectories [1 - 13]/15", "Emulating Trajectory 14/15",
        "Emulating Trajectory 15/15"]
    assert r.sample_final_state() == Counter(extra["reference_golden_counter"])
    # the three distinct SPAM trajectories (3-level register, each with its own term list) ran in ONE launch
    assert emu.last_engine_stats["n_launches"] == 1


def test_multi_level_trajectories_in_one_launch_equal_the_one_by_one_solves():
    """ryd_general_solve_many: the noise trajectories of a multi-level run (one general handle each) through
    one launch against the same handles solved one after the other."""
    from test_host_logic import _spam_all_emulator

    from pulser_amd.engine import GeneralEngine
    from pulser_amd.general import lower_general

    emu, _ = _spam_all_emulator()
    hd = emu._hamiltonian_data
    probs = [hd.problem(t, emu._sampling_rate) for t in hd.noise_trajectories]
    assert len(probs) >= 3
    times = np.asarray(emu._eval_times_array)
    psi0 = np.asarray(emu._initial_state).reshape(-1)
    engines = [GeneralEngine(lower_general(p, mesolve=False)) for p in probs]
    try:
        st_a = [e.new_state(psi0) for e in engines]
        many = [s.cpu().numpy() for s in GeneralEngine.solve_many(engines, st_a, times)]
        assert engines[0].stats()["n_launches"] == 1 and all(e.stats()["n_launches"] == 0 for e in engines[1:])
        for e, got, sa in zip(engines, many, st_a):
            sb = e.new_state(psi0)
            one = e.solve(sb, times).cpu().numpy()
            assert np.array_equal(got, one)  # same kernel body, same schedule: bit-identical
            assert np.array_equal(sa.cpu().numpy(), sb.cpu().numpy())
    finally:
        for e in engines:
            e.close()


def test_reference_results_noisy_goldens_end_to_end():
    """test_simresults.py:63-90, 383-389, 446-456 with the real solver on the GPU:
    15 doppler/amplitude/SPAM trajectories x 1001 evaluation times, seed 123."""
    from test_host_logic import _check_results_noisy, _results_noisy_emulator

    emu, extra = _results_noisy_emulator()
    with pytest.warns(DeprecationWarning):
        r = emu.run()
    assert dict(r[-1].bitstring_counts) == extra["oracle_total_final_counter"]
    _check_results_noisy(r, extra)
    # test_simresults.py:400-409 (the same sequence without noise)
    from pulser_amd.hamiltonian_data import SequenceInputs

    clean = QutipEmulator(SequenceInputs.from_dict(load_fixture("results_noisy.npz")[0]["inputs"]))
    with pytest.warns(DeprecationWarning):
        res = clean.run()
    np.random.seed(123)
    assert res.sample_final_state(1) == Counter({"11": 1})


def test_reference_get_final_state_noisy_golden_end_to_end():
    """test_simresults.py:244-275 with the real solver on the GPU (digital basis,
    doppler + trap position fluctuations + SPAM, seed 123)."""
    from test_host_logic import _check_final_state_noisy, _final_state_noisy_emulator

    emu, extra = _final_state_noisy_emulator()
    with pytest.warns(DeprecationWarning):
        r = emu.run()
    _check_final_state_noisy(r, extra)


@pytest.mark.parametrize("ch", ["mw_global", "rydberg_global", "raman_global"])
def test_slm_effective_size_hamiltonians(ch):
    """test_simulation.py:1928-2000: H(0) of the first SPAM trajectory with an SLM
    mask (XY: only the unmasked, well-prepared atom3 is driven: amp/2 sigma_x)."""
    from test_host_logic import _slm_effective_size_emulator

    emu, extra = _slm_effective_size_emulator(ch)
    h0 = np.asarray(emu.get_hamiltonian(0))
    np.testing.assert_allclose(h0, extra["oracle_h0"], atol=1e-9 * np.abs(extra["oracle_h0"]).max())
    if ch == "mw_global":
        sx = np.array([[0.0, 1.0], [1.0, 0.0]])
        np.testing.assert_allclose(h0, 0.5 * np.kron(np.eye(8), sx), atol=1e-12)


def test_slm_mask_xy_equals_removing_the_qubit():
    """test_simulation.py:1748-1838 through ``get_hamiltonian`` on the GPU: while the
    mask is on, H_masked = H_two x 1; afterwards H_masked = H_three."""
    from test_host_logic import _slm_mask_emulators

    (masked, three, two, eq_m, eq_2), extra = _slm_mask_emulators(
        "tp_masked", "tp_three", "tp_two", "eq_masked", "eq_two")
    ti, tf = (int(x) for x in extra["tp_mask_time"])
    for t in (0, 37, 99, 101, 150, 299):  # sample 100 itself is the switch-over knot
        m = np.asarray(masked.get_hamiltonian(t))
        if ti <= t < tf:
            np.testing.assert_allclose(m, np.kron(np.asarray(two.get_hamiltonian(t)), np.eye(2)), atol=1e-10)
        else:
            np.testing.assert_allclose(m, np.asarray(three.get_hamiltonian(t)), atol=1e-10)
    for t in (0, 50, 99):
        np.testing.assert_allclose(np.asarray(eq_m.get_hamiltonian(t)),
                                   np.kron(np.asarray(eq_2.get_hamiltonian(t)), np.eye(2)), atol=1e-10)


@pytest.mark.parametrize("k", [0, 1])
def test_reference_golden_counters_eom_detuning_limits_end_to_end(k):
    """test_simulation.py:2593-2660 with the real solver: |detuning| = 1000 rad/us
    for 4.5 us (the stiffest golden), then the detuning_sigma variant must run."""
    from test_host_logic import _eom_emulator
    from pulser_amd.hamiltonian_data import SequenceInputs

    emu, extra = _eom_emulator(k)
    with pytest.warns(DeprecationWarning):
        res = emu.run()
    assert res.sample_final_state() == Counter(extra["reference_golden_counter"])
    final = np.asarray(res.states[-1]).ravel()
    assert np.max(np.abs(final - extra["oracle_final_state_tight"])) < 1e-7
    noisy = QutipEmulator(SequenceInputs.from_dict(load_fixture(f"eom_limit_det_{k}.npz")[0]["inputs"]),
                          noise_model=NoiseModel(detuning_sigma=0.1), n_trajectories=1)
    with pytest.warns(DeprecationWarning):
        r = noisy.run()
    assert sum(sum(x.bitstring_counts.values()) for x in r) == len(r) == len(extra["eval_times"])


def test_relaxation_noise_population_decays(capsys):
    """test_simulation.py:1049-1076: one atom, Blackman pi pulse then 10 us of free
    relaxation (rate 0.1): the sampled Rydberg population decays monotonically."""
    w = np.clip(np.blackman(1000), 0, np.inf)
    amp = np.concatenate((w * np.pi / (w.sum() * 1e-3), np.zeros(10000), [0.0]))
    inputs = single_global_channel(np.zeros((1, 2)), dict(amp=amp, det=0 * amp, phase=0 * amp),
                                   P.C6_LEVEL70)
    emu = QutipEmulator(inputs, noise_model=NoiseModel(relaxation_rate=0.1))
    assert len(emu._current_problem["collapse_ops"]) == 1
    with pytest.warns(DeprecationWarning):
        res = emu.run(print_progress=True)
    assert capsys.readouterr().out.rstrip("\n").split("\n") == ["Emulating Trajectory 1/1"]
    np.random.seed(7)
    start = res.sample_state(1)
    pop = start["1"]
    assert pop > start.get("0", 0)
    for t in range(2, 10):
        new = res.sample_state(t)["1"]
        assert new < pop
        pop = new
    rho = np.asarray(res.get_state(6.0))
    assert abs(rho[0, 0].real - np.asarray(res.get_state(1.0))[0, 0].real * np.exp(-0.5)) < 1e-6


def _blackman(duration, area):
    w = np.clip(np.blackman(duration), 0, np.inf)
    return w * area / (w.sum() * 1e-3)


def test_pulses_between_long_delays_are_not_skipped():
    """test_simulation.py:612-633: delay, pi pulse, delay, pi/2 pulse on one atom ends
    at population 1/2.  (The reference's other half - with ``max_step=1`` QuTiP's
    adaptive integrator steps over both pulses and returns 0 - is an artefact this
    engine does not reproduce: a larger ``max_step`` never skips samples here.)"""
    amp = np.concatenate((np.zeros(1500), _blackman(600, np.pi), np.zeros(2000), _blackman(600, np.pi / 2)))
    inputs = single_global_channel(np.zeros((1, 2)), dict(amp=amp, det=0 * amp, phase=0 * amp),
                                   P.C6_LEVEL70, extended=False)
    emu = QutipEmulator(inputs)
    proj = np.diag([1.0, 0.0]).astype(complex)
    for kw in ({}, {"max_step": 1}):
        with pytest.warns(DeprecationWarning):
            occ = emu.run(**kw).expect([proj])[0]
        assert np.isclose(occ[-1], 0.5, 1e-4)
        assert np.isclose(occ[2100], 1.0, 1e-6)


def test_single_atom_and_empty_sequences():
    """test_simulation.py:434-472, 591-609."""
    from pulser_amd.hamiltonian_data import ChannelInput, SequenceInputs, Slot

    amp = np.ones(16)
    one = single_global_channel(np.zeros((1, 2)), dict(amp=amp, det=amp, phase=0 * amp), P.C6_LEVEL70,
                                extended=False)
    for ev in ("Full", "Minimal"):
        with pytest.warns(DeprecationWarning):
            res = QutipEmulator(one, evaluation_times=ev).run()
        assert res._size == 1 and len(res) == (17 if ev == "Full" else 2)
    empty = SequenceInputs(np.zeros((1, 2)), ("q0",), [ChannelInput("ch0", "Global", "XY", np.zeros(0),
                                                                    np.zeros(0), np.zeros(0))], 1.0)
    with pytest.raises(ValueError, match="SequenceSamples is empty"):
        QutipEmulator(empty)
    # only delays + SPAM: every sample of every atom is zero, the run still works
    coords = np.array([[-4.0, 0.0], [0.0, 4.0], [4.0, 0.0]])
    z = np.zeros(100)
    idle = SequenceInputs(coords, ("control1", "target", "control2"),
                          [ChannelInput("test", "Local", "digital", z, z, z, slots=[Slot(0, 100, (1,))]),
                           ChannelInput("test2", "Global", "ground-rydberg", z, z, z)], P.C6_LEVEL70)
    np.random.seed(1)
    emu = QutipEmulator(idle, noise_model=NoiseModel(samples_per_run=1, state_prep_error=0.005,
                                                     p_false_pos=0.01, p_false_neg=0.05), n_trajectories=15)
    nested = emu._current_problem["samples"]
    assert not nested["Global"]
    for per_atom in nested["Local"].values():
        for entry in per_atom.values():
            assert not any(np.any(v) for v in entry.values())
    with pytest.warns(DeprecationWarning):
        r = emu.run()
    assert sum(r[-1].bitstring_counts.values()) == 15


def test_sharded_ensemble_on_the_gpu_equals_the_serial_run():
    """pulser_amd.distributed.run_ensemble with the real HIP solver (world size 1):
    same Counters as QutipEmulator.run() for the same seed (all random numbers are
    drawn in the reference's order), independent of the batch size; with
    dissipation the quantum-jump seeds are per trajectory, so the split does not
    matter either."""
    from pulser_amd.distributed import run_ensemble

    prob0, extra = load_fixture("cfg4_chain12_noise.npz")
    nm = NoiseModel(**extra["noise_model"])

    def make(noise=nm, n=24):
        np.random.seed(7)
        return QutipEmulator(_chain12_inputs(extra), noise_model=noise, n_trajectories=n,
                             evaluation_times="Minimal")

    with pytest.warns(DeprecationWarning):
        serial = make().run()
    for batch in (5, 64):
        out = run_ensemble(make(), dist=None, batch=batch)
        assert out["counters"][-1] == Counter(serial[-1].bitstring_counts)
        assert out["n_measures"] == serial.n_measures
    # dissipative + stochastic noise -> Monte-Carlo wavefunction trajectories
    nm_mc = NoiseModel(temperature=50.0, amp_sigma=0.05, dephasing_rate=0.2, relaxation_rate=0.1)
    a = run_ensemble(make(nm_mc, 12), dist=None, batch=4, mc_seed=11)
    b = run_ensemble(make(nm_mc, 12), dist=None, batch=12, mc_seed=11)
    assert a["counters"] == b["counters"]
    assert np.array_equal(a["histograms"], b["histograms"])
    c = run_ensemble(make(nm_mc, 12), dist=None, batch=12, mc_seed=12)
    assert sum(c["counters"][-1].values()) == sum(a["counters"][-1].values())


@pytest.mark.parametrize("generic", [False, True])
def test_hf_detuning_noise_factored_on_device_equals_per_trajectory_solve(generic):
    """High-frequency detuning noise + laser-waist amplitude noise + doppler +
    register noise (pulser-core's trajectories, fixture): the factored lowering
    with the on-device ``ryd_dterm`` synthesis gives the states of the
    per-trajectory lowering, on the persistent and on the multi-launch kernels."""
    from pulser_amd.engine import Engine
    from pulser_amd.hamiltonian_data import HamiltonianData, SequenceInputs
    from pulser_amd.terms import lower

    prob, extra = load_fixture("waist_tri6.npz")
    inputs = SequenceInputs.from_dict(prob["inputs"])
    kw = dict(extra["noise_model"])
    for key in ("detuning_hf_psd", "detuning_hf_omegas"):
        kw[key] = tuple(kw[key])
    np.random.seed(5)
    hd = HamiltonianData(inputs.extend_duration(inputs.max_duration + 1), NoiseModel(**kw), 4)
    trajs = hd.noise_trajectories
    times = [0.0, 0.2, inputs.max_duration * 1e-3]
    out = []
    for tables in (hd.device_tables(trajs, 1.0), lower([hd.problem(t, 1.0) for t in trajs])):
        with Engine(tables, mode="sesolve") as eng:
            eng.set_path(generic)
            state = eng.new_state()
            out.append(eng.solve(state, times).cpu().numpy())
            launches = eng.stats()["n_launches"]
    assert (launches == 1) == (not generic)
    assert np.max(np.abs(out[0] - out[1])) < 1e-9
    assert np.max(np.abs(out[0][-1][0] - out[0][-1][1])) > 1e-3  # trajectories really differ


@pytest.mark.parametrize("fixture,mesolve", [("noises_all_0.npz", True), ("noises_all_0.npz", False),
                                             ("noisy_xy_0.npz", True), ("noises_digital_6.npz", True)])
def test_general_path_persistent_kernel_matches_multi_launch(fixture, mesolve):
    """Explicit-term engine (3- / 4-level bases, XY): the one-launch kernel for
    vectors of at most 4096 entries against one launch per Taylor stage."""
    from pulser_amd.engine import GeneralEngine
    from pulser_amd.general import lower_general
    from pulser_amd.hamiltonian_data import SequenceInputs

    prob, extra = load_fixture(fixture)
    if "inputs" in prob:  # XY fixtures store the sequence inputs
        emu = QutipEmulator(SequenceInputs.from_dict(prob["inputs"]), sampling_rate=0.1,
                            noise_model=NoiseModel(dephasing_rate=0.05))
        prob = emu._current_problem
        init = np.asarray(emu.initial_state).reshape(-1)
    else:
        d, n = len(prob["eigenbasis"]), prob["n_qudits"]
        init = np.zeros(d**n, dtype=complex)
        init[-1 if d == 2 else sum((list(prob["eigenbasis"]).index("g")) * d**k for k in range(n))] = 1.0
    tables = lower_general(prob, mesolve=mesolve)
    T = int(prob["duration"]) - 1
    times = np.array([0.0, 0.3 * T * 1e-3, 0.3 * T * 1e-3, T * 1e-3])
    outs, launches = [], []
    for multi in (False, True):
        with GeneralEngine(tables) as eng:
            eng.set_path(multi)
            st = eng.new_state(init)
            outs.append(eng.solve(st, times).cpu().numpy())
            launches.append(eng.stats()["n_launches"])
    assert launches[0] == 2 and launches[1] > 20
    assert np.max(np.abs(outs[0] - outs[1])) < 1e-12
    assert np.max(np.abs(outs[0][-1] - outs[0][0])) > 1e-3
