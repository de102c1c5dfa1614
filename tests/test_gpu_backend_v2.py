"""GPU: the QutipBackendV2-style backend (observables engine) against oracle
values, following tests/pulser_simulation/test_qutip_backend_v2.py."""
from collections import Counter

import numpy as np
import pytest

from helpers import blockade_radius

from pulser_amd import NoiseModel, Solver
from pulser_amd import problem as P
from pulser_amd.backend import (BitStrings, CorrelationMatrix, Energy, EnergySecondMoment,
                                EnergyVariance, Fidelity, Occupation, QutipBackendV2, QutipConfig,
                                Results, RydState, StateResult)
from pulser_amd.hamiltonian_data import single_global_channel

pytestmark = pytest.mark.gpu


def _inputs(n=2):
    coords = P.register_coords(P.square_rect(1, n), blockade_radius())
    s = {k: v[:-1] for k, v in P.anneal_samples().items()}
    return single_global_channel(coords, s, P.C6_LEVEL70, extended=False), coords


def _oracle(n, times):
    from oracle import qutip_path as qp

    coords = P.register_coords(P.square_rect(1, n), blockade_radius())
    prob = P.make_ising_problem(coords, P.anneal_samples())
    ham = qp.build_hamiltonian(prob)
    states = qp.sesolve(ham, qp.all_ground_state(n, prob["eigenbasis"]), times, max_step=1e-3, **qp.TIGHT)
    return ham, states


def test_backend_v2_observables_against_oracle(capfd):
    """test_qutip_backend_v2.py:112-154 plus every default observable."""
    from oracle import sampling as osamp

    n = 3
    inputs, _ = _inputs(n)
    target = RydState.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"rgr": 1.0})
    rel = [0.0, 0.25, 0.5, 1.0]
    config = QutipConfig(
        default_evaluation_times=rel,
        observables=[StateResult(), BitStrings(num_shots=200), Occupation(), CorrelationMatrix(),
                     Energy(evaluation_times=[0.0, 0.5, 1.0]), EnergyVariance(), EnergySecondMoment(),
                     Fidelity(target)],
        print_progress=True,
    )
    with pytest.raises(TypeError, match="'config' must be an instance of 'EmulationConfig'"):
        QutipBackendV2(inputs, config="tralala")
    backend = QutipBackendV2(inputs, config=config)
    np.random.seed(77)
    results = backend.run()
    out, _ = capfd.readouterr()
    assert out == "Emulating Trajectory 1/1\n"
    assert set(results.get_result_tags()) == {"state", "bitstrings", "occupation", "correlation_matrix",
                                              "energy", "energy_variance", "energy_second_moment", "fidelity"}
    assert results.get_result_times("state") == rel
    assert results.get_result_times("energy") == [0.0, 0.5, 1.0]
    times = np.array(rel) * 3.1
    ham, ref = _oracle(n, times)
    np.random.seed(77)
    idx = np.arange(8)
    for i, t in enumerate(rel):
        psi = ref[i] / np.linalg.norm(ref[i])
        got = np.asarray(results.get_result("state", t).to_qobj())[:, 0]
        assert np.max(np.abs(got - psi)) < 1e-7
        assert results.get_result("bitstrings", t) == osamp.v2_sample(got, ("r", "g"), 200, "r")
        p = np.abs(psi) ** 2
        occ = [float(np.sum(p * (1 - ((idx >> (n - 1 - k)) & 1)))) for k in range(n)]
        assert np.allclose(results.get_result("occupation", t), occ, atol=1e-7)
        corr = np.array(results.get_result("correlation_matrix", t))
        assert np.allclose(np.diag(corr), occ, atol=1e-7) and np.allclose(corr, corr.T)
        assert abs(results.get_result("fidelity", t) - abs(psi[0b010]) ** 2) < 1e-7
        h = ham.matrix(times[i]).toarray()
        e2 = np.vdot(h @ psi, h @ psi).real
        e1 = np.vdot(psi, h @ psi).real
        assert abs(results.get_result("energy_second_moment", t) - e2) < 1e-6 * max(1, abs(e2))
        assert abs(results.get_result("energy_variance", t) - (e2 - e1 * e1)) < 1e-6 * max(1, abs(e2))
        if t in (0.0, 0.5, 1.0):
            assert abs(results.get_result("energy", t) - e1) < 1e-7 * max(1, abs(e1))
    assert results.get_result("energy", 0.0) == results.energy[0] == pytest.approx(0.0)
    assert results.final_bitstrings == results.bitstrings[-1]
    with pytest.raises(ValueError, match="not available at time"):
        results.get_result("energy", 0.25)


def test_backend_v2_default_config_and_noisy_aggregation():
    """Default config = final bitstrings + state; SPAM state-prep trajectories are
    aggregated: mean occupation, union of Counters, mean of |psi><psi|."""
    inputs, _ = _inputs(3)
    np.random.seed(5)
    res = QutipBackendV2(inputs).run()
    assert set(res.get_result_tags()) == {"bitstrings", "state"}
    assert sum(res.final_bitstrings.values()) == 1000
    nm = NoiseModel(state_prep_error=0.3, p_false_pos=0.02, p_false_neg=0.03)
    cfg = QutipConfig(noise_model=nm, n_trajectories=7,
                      observables=[BitStrings(evaluation_times=[1.0], num_shots=50), Occupation(),
                                   StateResult()])
    np.random.seed(9)
    backend = QutipBackendV2(inputs, config=cfg)
    trajs = backend._sim_obj._hamiltonian_data.noise_trajectories
    assert sum(t.reps for t in trajs) == 7 and len(trajs) > 1
    res = backend.run()
    assert sum(res.final_bitstrings.values()) == 7 * 50
    rho = np.asarray(res.state[-1].to_qobj())
    assert rho.shape == (8, 8) and abs(np.trace(rho) - 1) < 1e-9
    # aggregated occupation = trace of the aggregated density matrix with n_k
    idx = np.arange(8)
    occ = [float(np.sum(np.real(np.diag(rho)) * (1 - ((idx >> (2 - k)) & 1)))) for k in range(3)]
    assert np.allclose(res.occupation[-1], occ, atol=1e-9)
    assert isinstance(Results.aggregate([res]), Results)


def test_backend_v2_multilevel_and_xy_sequences():
    """3-level "all"-basis and XY sequences through the V2 backend: the states and
    the observables built on the noiseless H(t) (explicit-term engine on the GPU)
    equal the oracle's; occupation / energy identities hold."""
    from oracle import qutip_path as qp
    from helpers import load_fixture
    from test_host_logic import _inputs_from_problem
    from pulser_amd.hamiltonian_data import SequenceInputs

    cases = []
    prob, extra = load_fixture("noises_all_0.npz")
    meas = extra["aux"]["meas_basis"]
    cases.append((_inputs_from_problem(prob, measurement=meas if meas != "digital" else None), prob, False))
    xprob, _ = load_fixture("noisy_xy_0.npz")
    cases.append((SequenceInputs.from_dict(xprob["inputs"]), None, True))
    for inputs, prob, is_xy in cases:
        rel = [0.0, 0.5, 1.0]
        config = QutipConfig(default_evaluation_times=rel, sampling_rate=0.1,
                             observables=[StateResult(), Occupation(one_state=None if is_xy else "r"),
                                          Energy(), EnergySecondMoment(), EnergyVariance(),
                                          BitStrings(num_shots=50, one_state=None if is_xy else "r")])
        backend = QutipBackendV2(inputs, config=config)
        np.random.seed(3)
        res = backend.run()
        sim = backend._sim_obj
        noiseless = dict(sim._noiseless_problem)
        ham = qp.build_hamiltonian(noiseless)
        T = sim.total_duration_ns
        d = len(noiseless["eigenbasis"])
        assert d == (2 if is_xy else 3) and not sim._fast_path_ok(noiseless)
        for t in rel:
            st = res.get_result("state", t)
            psi = np.asarray(st.to_qobj())[:, 0]
            assert abs(np.linalg.norm(psi) - 1) < 1e-12
            H = ham.matrix(t * T / 1000).toarray()
            e = float(np.real(np.vdot(psi, H @ psi)))
            e2 = float(np.real(np.vdot(H @ psi, H @ psi)))
            assert abs(res.get_result("energy", t) - e) < 1e-9 * max(1.0, abs(e))
            assert abs(res.get_result("energy_second_moment", t) - e2) < 1e-9 * max(1.0, abs(e2))
            assert abs(res.get_result("energy_variance", t) - (e2 - e * e)) < 1e-7 * max(1.0, abs(e2))
            occ = res.get_result("occupation", t)
            n = st.n_qudits
            one = list(st.eigenstates).index("d" if is_xy else "r")
            idx = np.arange(d**n)
            ref = [float(np.sum(np.abs(psi[(idx // d ** (n - 1 - k)) % d == one]) ** 2)) for k in range(n)]
            assert np.allclose(occ, ref, atol=1e-12)
            assert sum(res.get_result("bitstrings", t).values()) == 50
        # final state against the oracle's tight integration of the same problem
        times = np.array([0.0, T * 1e-3])
        psi0 = qp.all_ground_state(noiseless["n_qudits"], noiseless["eigenbasis"], xy=is_xy)
        ref = qp.sesolve(ham, psi0, times, **qp.TIGHT)[-1]
        got = np.asarray(res.get_result("state", 1.0).to_qobj())[:, 0]
        assert np.max(np.abs(got - ref)) < 1e-7


def _constant_pulse_inputs(n, spacing, duration, amp, det=0.0):
    coords = np.stack([np.arange(n) * spacing, np.zeros(n)], axis=1)
    s = dict(amp=np.full(duration, amp), det=np.full(duration, det), phase=np.zeros(duration))
    return single_global_channel(coords, s, P.C6_LEVEL70, extended=False)


@pytest.mark.parametrize("amp_sigma", [0.0, 1.0])
def test_backend_v2_leakage_populations(amp_sigma):
    """tests/pulser_simulation/test_qutip_backend_v2.py:288-360: leakage |r> -> |x>,
    |g> -> |x> at equal rates on two far-apart atoms: the leaked populations follow
    1 - exp(-rate t) whatever the drive (and its amplitude noise) does."""
    rate, duration = 0.5, 500
    bx, bg, br = np.eye(3)[2][:, None], np.eye(3)[1][:, None], np.eye(3)[0][:, None]
    nm = NoiseModel(eff_noise_rates=[rate, rate], eff_noise_opers=[bx @ br.T, bx @ bg.T],
                    with_leakage=True, amp_sigma=amp_sigma)
    cfg = QutipConfig(default_evaluation_times=[1.0], observables=[StateResult(evaluation_times=[1.0])],
                      noise_model=nm, solver=Solver.MESOLVER, n_trajectories=1)
    np.random.seed(4)
    res = QutipBackendV2(_constant_pulse_inputs(2, 1000.0, duration, np.pi), config=cfg).run()
    rho = np.asarray(res.state[-1].to_qobj())
    assert rho.shape == (9, 9) and res.state[-1].eigenstates == ("r", "g", "x")
    px = bx @ bx.T
    keep = np.diag([1.0, 1.0, 0.0])
    e = np.exp(-rate * duration / 1000)
    ex = lambda op: float(np.real(np.trace(op @ rho)))  # noqa: E731
    assert ex(np.kron(px, keep) + np.kron(keep, px)) == pytest.approx(2 * (1 - e) * e)
    assert ex(np.kron(keep, keep)) == pytest.approx(e * e)
    assert ex(np.kron(px, px)) == pytest.approx((1 - e) ** 2)


def test_backend_v2_register_and_detuning_noise_give_a_mixed_state_and_fresh_draws():
    """test_qutip_backend_v2.py:363-393 (register + detuning noise -> averaged density
    matrix) and :558-580 (a second run draws new trajectories)."""
    nm = NoiseModel(trap_depth=1.0, trap_waist=1.0, temperature=50.0, disable_doppler=True,
                    detuning_sigma=5.0)
    assert set(nm.noise_types) == {"register", "detuning"}
    cfg = QutipConfig(default_evaluation_times=[1.0], observables=[StateResult(evaluation_times=[1.0])],
                      noise_model=nm, n_trajectories=10)
    np.random.seed(21)
    backend = QutipBackendV2(_constant_pulse_inputs(2, 1000.0, 500, np.pi), config=cfg)
    s1 = np.asarray(backend.run().state[-1].to_qobj())
    s2 = np.asarray(backend.run().state[-1].to_qobj())
    assert s1.shape == (4, 4) and abs(np.trace(s1) - 1) < 1e-9
    assert np.trace(s1 @ s1).real < 1 - 1e-6  # mixed
    overlap = np.trace(s1.conj().T @ s2) / (np.linalg.norm(s1) * np.linalg.norm(s2))
    assert overlap != pytest.approx(1.0)


def test_backend_v2_evaluation_time_rounding():
    """test_qutip_backend_v2.py:257-285 (100 relative times must give 100 states for
    durations that do not divide nicely) and :469-492 (a time that differs from a
    grid point by one rounding error is not duplicated)."""
    times = np.linspace(0, 1, 100).tolist()
    for duration in (400, 428, 472, 516, 596):
        coords = np.array([[-5.0, 0.0], [5.0, 0.0]])
        s = dict(amp=np.full(duration, np.pi), det=np.zeros(duration), phase=np.zeros(duration))
        inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
        res = QutipBackendV2(inputs, config=QutipConfig(observables=[StateResult(evaluation_times=times)])).run()
        assert len(res.state) == 100
    inputs = _constant_pulse_inputs(1, 1.0, 1000, 1.0)
    cfg = QutipConfig(observables=[BitStrings(evaluation_times=np.linspace(0.0, 1.0, 1001)),
                                   BitStrings(evaluation_times=[0.49299999999999994], tag_suffix="mod")])
    res = QutipBackendV2(inputs, config=cfg).run()
    # like the reference's test, the point is that this runs (no "value already stored"
    # error); both 0.493 and 0.49299999999999994 are evaluation times of the run
    assert len(res.bitstrings) >= 1001 and len(res.bitstrings_mod) >= 1


def test_backend_v2_run_from_sequence_samples_is_the_same_run():
    """test_qutip_backend_v2.py:616-655: running from already sampled sequences gives
    bit-identical states."""
    inputs = _constant_pulse_inputs(1, 1.0, 1000, 1.0)
    cfg = QutipConfig(observables=[StateResult()],
                      initial_state=RydState.from_state_amplitudes(eigenstates=("r", "g"),
                                                                   amplitudes={"g": 1.0}))
    for config in (None, cfg):
        s1 = np.asarray(QutipBackendV2(inputs, config=config).run().state[-1].to_qobj())
        s2 = np.asarray(QutipBackendV2.run_from_sequence_samples(inputs, config=config).state[-1].to_qobj())
        assert np.array_equal(s1, s2)


def test_legacy_qutip_backend():
    """tests/pulser_simulation/test_qutip_backend.py:42-100: the deprecated wrapper
    (one atom, local Raman pi pulse -> |h>; SPAM noise -> NoisyResults)."""
    from pulser_amd import EmulatorConfig, QutipBackend
    from pulser_amd.hamiltonian_data import ChannelInput, SequenceInputs, Slot
    from pulser_amd.results import CoherentResults, NoisyResults

    w = np.clip(np.blackman(1000), 0, np.inf)
    amp = w * np.pi / (w.sum() * 1e-3)
    inputs = SequenceInputs(np.zeros((1, 2)), ("q0",),
                            [ChannelInput("raman_local", "Local", "digital", amp, 0 * amp, 0 * amp,
                                          slots=[Slot(0, 1000, (0,))])], P.C6_LEVEL70)
    with pytest.raises(TypeError, match="must be of type 'EmulatorConfig'"), pytest.deprecated_call():
        QutipBackend(inputs, NoiseModel())
    with pytest.deprecated_call(match="'QutipBackend' is deprecated"):
        backend = QutipBackend(inputs)
    results = backend.run()
    assert isinstance(results, CoherentResults)
    assert np.array_equal(np.asarray(results[0].get_state()).ravel(), [1, 0])  # |g> in (g, h)
    final = np.asarray(results[-1].get_state())
    assert np.array_equal(final, np.asarray(results.get_final_state()))
    np.testing.assert_allclose(np.abs(final), [[0], [1]], atol=1e-5)
    spam = NoiseModel(p_false_pos=0.1, p_false_neg=0.05, state_prep_error=0.1, runs=10, samples_per_run=1)
    with pytest.deprecated_call():
        backend = QutipBackend(inputs, config=EmulatorConfig(noise_model=spam, evaluation_times="Minimal"))
    np.random.seed(2)
    assert isinstance(backend.run(), NoisyResults) and backend._sim_obj.noise_model == spam
    with pytest.raises(ValueError, match="'evaluation_times' must be one of the following options"):
        EmulatorConfig(evaluation_times="Best")


def test_backend_v2_callbacks_and_device_noise_model(capfd):
    """tests/pulser_simulation/test_qutip_backend_v2.py:91-108 (a callback sees every
    evaluation time of a "Full" run, noisy or not) and :158-196 (the device's noise model
    wins when the configuration prefers it; the configuration keeps its own)."""
    class CountCalls:
        def __init__(self):
            self.counter = 0

        def __call__(self, config, t, state, hamiltonian, result):
            self.counter += 1

    inputs = _constant_pulse_inputs(2, 8.0, 120, np.pi)
    for kw in ({}, {"noise_model": NoiseModel(amp_sigma=0.1), "n_trajectories": 1}):
        backend = QutipBackendV2(inputs, config=QutipConfig(callbacks=[CountCalls()], **kw))
        np.random.seed(3)
        backend.run()
        assert backend._config.callbacks[0].counter == 120 + 1
    with pytest.raises(TypeError, match="'config' must be an instance of 'EmulationConfig'"):
        QutipBackendV2(inputs, config="tralala")

    class Device:
        noise_model = NoiseModel(dephasing_rate=0.01, temperature=50.0)

    class Holder:  # a sequence-like object that only carries what the backend reads
        device = Device()
        channels = inputs.channels

    cfg = QutipConfig(observables=[StateResult(evaluation_times=[1.0])],
                      noise_model=NoiseModel(p_false_neg=0.1), prefer_device_noise_model=True,
                      n_trajectories=2, print_progress=True)
    assert QutipBackendV2._get_noise_model(cfg, Holder.device) is Device.noise_model
    assert cfg.noise_model.p_false_neg == 0.1
    sim_cfg = cfg.with_changes(prefer_device_noise_model=False, noise_model=Device.noise_model)
    capfd.readouterr()
    np.random.seed(4)
    QutipBackendV2(inputs, config=sim_cfg).run()
    assert capfd.readouterr().out == "Emulating Trajectory 1/2\nEmulating Trajectory 2/2\n"


def test_device_side_observables_one_call_per_evaluation_time():
    """Occupation / CorrelationMatrix / Energy / EnergyVariance /
    EnergySecondMoment of a 12-atom sequence at 100 evaluation times come from ONE ``ryd_observe``
    call per time (pair reduction + one generator application + one dot), not from per-observable
    expectation values or a materialised H(t); values against NumPy on the stored states."""
    from oracle import qutip_path as qp
    from pulser_amd import problem as P
    from pulser_amd.backend import EnergySecondMoment, EnergyVariance
    from pulser_amd.hamiltonian_data import single_global_channel

    n = 12
    coords = P.register_coords(P.square_rect(1, n), blockade_radius())
    smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
    inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
    times = np.linspace(0.01, 1.0, 100).tolist()
    obs = [StateResult(evaluation_times=[0.25, 1.0]), Occupation(), CorrelationMatrix(), Energy(),
           EnergyVariance(), EnergySecondMoment()]
    res = QutipBackendV2(inputs, config=QutipConfig(default_evaluation_times=times, observables=obs)).run()
    stats = QutipBackendV2.last_observable_engine_stats
    # per evaluation time: pair reduction + generator application + energy dot = 3 launches
    assert stats["n_launches"] == 3 * len(times) and stats["n_applications"] == len(times)
    timing = QutipBackendV2.last_timing
    assert timing["observables_s"] < 0.5 * timing["solve_s"] + 0.5, timing
    prob = P.make_ising_problem(coords, P.anneal_samples())
    ham = qp.build_hamiltonian(prob)
    idx = np.arange(1 << n)
    nk = np.stack([1 - ((idx >> (n - 1 - k)) & 1) for k in range(n)], axis=1).astype(float)
    for t in (0.25, 1.0):
        psi = np.asarray(res.get_result(obs[0], t).to_qobj())[:, 0]
        psi = psi / np.linalg.norm(psi)
        p = np.abs(psi) ** 2
        assert np.allclose(res.get_result(obs[1], t), p @ nk, atol=1e-12)
        assert np.allclose(res.get_result(obs[2], t), (nk * p[:, None]).T @ nk, atol=1e-12)
        h_psi = ham.apply(t * 3.1, psi)
        e, e2 = np.vdot(psi, h_psi).real, np.vdot(h_psi, h_psi).real
        assert abs(res.get_result(obs[3], t) - e) < 1e-9 * max(1.0, abs(e))
        assert abs(res.get_result(obs[5], t) - e2) < 1e-9 * max(1.0, e2)
        assert abs(res.get_result(obs[4], t) - (e2 - e * e)) < 1e-7 * max(1.0, e2)


def test_device_side_observables_of_density_matrices():
    """Master-equation run (dephasing, no stochastic noise -> mesolve): the observables of the
    density matrices come from the same one-call device path - occupations / correlations from the
    diagonal, Tr(H rho) and Tr(H^2 rho) from the elements within two bit flips of it - against
    dense NumPy with the oracle's H(t)."""
    from oracle import qutip_path as qp
    from pulser_amd import problem as P
    from pulser_amd.backend import EnergySecondMoment, EnergyVariance
    from pulser_amd.hamiltonian_data import single_global_channel

    n = 6
    coords = P.register_coords(P.triangular_rect(2, 3), blockade_radius())
    smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
    inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
    times = [0.2, 0.6, 1.0]
    obs = [StateResult(), Occupation(), CorrelationMatrix(), Energy(), EnergyVariance(), EnergySecondMoment()]
    cfg = QutipConfig(default_evaluation_times=times, observables=obs, noise_model=NoiseModel(dephasing_rate=0.3))
    res = QutipBackendV2(inputs, config=cfg).run()
    stats = QutipBackendV2.last_observable_engine_stats
    assert stats["n_launches"] == 2 * len(times)  # pair reduction + the Tr(H rho) gather per time
    ham = qp.build_hamiltonian(P.make_ising_problem(coords, P.anneal_samples()))
    idx = np.arange(1 << n)
    nk = np.stack([1 - ((idx >> (n - 1 - k)) & 1) for k in range(n)], axis=1).astype(float)
    for t in times:
        rho = np.asarray(res.get_result(obs[0], t).to_qobj())
        assert rho.shape == (64, 64)
        rho = rho / np.trace(rho).real
        p = np.real(np.diag(rho))
        H = ham.matrix(t * 3.1).toarray()
        e, e2 = np.trace(H @ rho).real, np.trace(H @ H @ rho).real
        assert np.allclose(res.get_result(obs[1], t), p @ nk, atol=1e-12)
        assert np.allclose(res.get_result(obs[2], t), (nk * p[:, None]).T @ nk, atol=1e-12)
        assert abs(res.get_result(obs[3], t) - e) < 1e-9 * max(1.0, abs(e))
        assert abs(res.get_result(obs[5], t) - e2) < 1e-9 * max(1.0, e2)
        assert abs(res.get_result(obs[4], t) - (e2 - e * e)) < 1e-7 * max(1.0, e2)
