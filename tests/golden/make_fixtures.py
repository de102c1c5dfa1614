"""Generate the golden fixtures under tests/golden/ (run in the BUILD container).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=/tmp/shim:/root/reference/pulser-core:/root/repo \
        python /root/repo/tests/golden/make_fixtures.py [name ...]

Names (default: rydberg digital three cfg1 cfg2 cfg3 cfg4): rydberg, digital, xy, all, three, cfg1..cfg4,
spam_all, results_noisy, final_state_noisy, slm_effective_size, slm_masks, modulation, eom_limit_det,
multichannel_noise, dmm, results, waist, config, ns14 (14-atom headline, tight), cfg3_8 / cfg3_10 / cfg3_12
(interacting 8- / 10- / 12-atom Lindblad, tight; cfg3_12_to1300: the first 1.3 us of the 12-atom one), rect16 (16-atom square register, tight).

* Inputs are captured by importing the reference's ``pulser-core`` (read-only,
  never shipped; needs the no-op ``jsonschema``/``referencing`` stand-in of
  SURVEY.md Appendix B1 in /tmp/shim) and replaying the constructor order of
  ``QutipEmulator`` (pulser-simulation/pulser_simulation/simulation.py:130-230)
  so that the global ``np.random`` stream is consumed exactly as the reference
  does.
* Expected outputs are (a) the literal golden values of the reference's own
  tests (known answers, cited per fixture) and (b) outputs of the CPU oracle
  (``oracle/``) at QuTiP-default and at tight tolerances.

Fixtures are data only: arrays + JSON metadata (``pulser_amd.problem.save_problem``).
"""

from __future__ import annotations

import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import pulser  # noqa: E402
from pulser import NoiseModel, Pulse, Register, Sequence  # noqa: E402
from pulser._hamiltonian_data import HamiltonianData  # noqa: E402
from pulser.devices import AnalogDevice, DigitalAnalogDevice, MockDevice  # noqa: E402
from pulser.noise_model import _LEGACY_DEFAULTS  # noqa: E402
from pulser.sampler import sampler  # noqa: E402
from pulser.waveforms import BlackmanWaveform, RampWaveform  # noqa: E402

from oracle import qutip_path as qp  # noqa: E402
from oracle import sampling as osamp  # noqa: E402
from pulser_amd import problem as P  # noqa: E402
from pulser_amd.pulser_adapter import (  # noqa: E402
    channel_amp_det,
    problems_from_hamiltonian_data,
)

warnings.simplefilter("ignore")


def capture(seq, noise_model, sampling_rate=1.0, n_trajectories=None):
    """Replay ``QutipEmulator.from_sequence`` up to the solver call."""
    samples = sampler.sample(seq, extended_duration=seq.get_duration())
    T = samples.max_duration
    ext = samples.extend_duration(T + 1)  # simulation.py:173
    nm = noise_model or NoiseModel()
    ntraj = n_trajectories if n_trajectories is not None else nm.runs
    hd = HamiltonianData(ext, seq.register, seq.device, nm, ntraj)  # :208
    problems = list(problems_from_hamiltonian_data(hd, sampling_rate))
    # set_evaluation_times("Full") -> hidden noiseless HamiltonianData (:266-297)
    HamiltonianData(ext, seq.register, seq.device, NoiseModel(), 1)
    opts = qp.default_options(channel_amp_det(ext), T)
    tlist = qp.sampling_times(T + 1, sampling_rate)
    eval_times = np.union1d(tlist, [0.0, T * 1e-3])  # :596-598
    meas = ext._measurement
    bname = problems[0]["basis_name"]
    if not meas:
        meas = "digital" if "all" in bname else bname.replace("_with_error", "")
    aux = {
        "T": int(T),
        "eval_times": eval_times,
        "options": {k: float(v) for k, v in opts.items()},
        "meas_basis": meas,
        "matching_meas_basis": bool(meas in bname),
        "channel_amp_det": [np.stack(ad) for ad in channel_amp_det(ext)],
    }
    return problems, aux, hd


def solve(problem, aux, psi0=None, tight=False):
    ham = qp.build_hamiltonian(problem)
    if psi0 is None:
        psi0 = qp.all_ground_state(problem["n_qudits"], problem["eigenbasis"])
    opts = dict(aux["options"])
    if tight:
        opts.update(qp.TIGHT)
    fn = qp.mesolve if problem["collapse_ops"] else qp.sesolve
    return fn(ham, psi0, aux["eval_times"], **opts)


# ---------------------------------------------------------------------------
# 1. test_simulation.py:978-1040  (1 atom, 7 noise combos, seed 123)
# ---------------------------------------------------------------------------

Z2 = np.array([[1, 0], [0, -1]], dtype=complex)

RYDBERG_CASES = [
    (("dephasing",), {"0": 572, "1": 428}, 1),
    (("relaxation",), {"0": 572, "1": 428}, 1),
    (("eff_noise",), {"0": 572, "1": 428}, 1),
    (("depolarizing",), {"0": 561, "1": 439}, 3),
    (("dephasing", "depolarizing", "relaxation"), {"0": 562, "1": 438}, 5),
    (("eff_noise", "dephasing"), {"0": 573, "1": 427}, 2),
    (("eff_noise", "leakage"), {"0": 572, "1": 428}, 1),
]


def legacy_params(noise):
    params = {
        p: _LEGACY_DEFAULTS[p]
        for p in NoiseModel._find_relevant_params(
            [n for n in noise if n not in ["leakage", "eff_noise"]],
            state_prep_error=_LEGACY_DEFAULTS["state_prep_error"],
            amp_sigma=_LEGACY_DEFAULTS["amp_sigma"],
            laser_waist=_LEGACY_DEFAULTS["laser_waist"],
        )
    }
    return params


def gen_noises_rydberg():
    for k, (noise, golden, n_ops) in enumerate(RYDBERG_CASES):
        np.random.seed(123)
        reg = Register.from_coordinates([(0, 0)], prefix="q")
        seq = Sequence(reg, DigitalAnalogDevice)
        seq.declare_channel("ch0", "rydberg_global")
        seq.add(Pulse.ConstantPulse(2500, np.pi, 0, 0), "ch0")
        params = legacy_params(noise)
        with_leakage = "leakage" in noise
        if with_leakage or "eff_noise" in noise:
            params["eff_noise_opers"] = [
                np.array([[1, 0, 0], [0, 0, 0], [0, 0, 0]], dtype=complex)
                if with_leakage
                else Z2
            ]
            params["eff_noise_rates"] = [0.1 if with_leakage else 0.025]
        ntraj = params.pop("runs", None)
        problems, aux, _ = capture(
            seq, NoiseModel(with_leakage=with_leakage, **params), 0.01, ntraj
        )
        assert len(problems) == 1
        rng_state_after_ctor = np.random.get_state()[1][:4].copy()
        states = solve(problems[0], aux)
        counter = osamp.sample_state(
            states, aux["eval_times"], aux["eval_times"][-1], 1000,
            problems[0]["n_qudits"], problems[0]["eigenbasis"],
            aux["meas_basis"], aux["matching_meas_basis"],
        )
        ok = dict(counter) == golden
        print(f"noises_rydberg[{k}] {noise}: oracle {dict(counter)} golden {golden} {'OK' if ok else 'MISMATCH'}")
        tight = solve(problems[0], aux, tight=True)
        P.save_problem(
            os.path.join(HERE, f"noises_rydberg_{k}.npz"),
            problems[0],
            aux=aux,
            seed=123,
            noise=list(noise),
            n_collapse_ops=n_ops,
            reference_golden_counter=golden,
            reference_cite="tests/pulser_simulation/test_simulation.py:978-1040",
            oracle_final_state_default=states[-1],
            oracle_final_state_tight=tight[-1],
            oracle_lookup_index=osamp.index_from_time(aux["eval_times"], aux["eval_times"][-1]),
            oracle_lookup_state_default=states[osamp.index_from_time(aux["eval_times"], aux["eval_times"][-1])],
            rng_probe_after_ctor=rng_state_after_ctor,
        )


# ---------------------------------------------------------------------------
# 2. test_simulation.py:1079-1160 (3 atoms, digital basis, local pulses)
# ---------------------------------------------------------------------------

deph_res = {"111": 978, "110": 12, "011": 7, "101": 3}
depo_res = {"111": 827, "101": 63, "011": 59, "110": 40, "010": 5, "001": 4, "000": 1, "100": 1}
deph_depo_res = {"111": 807, "101": 64, "011": 60, "110": 56, "001": 5, "010": 4, "100": 3, "000": 1}
eff_deph_res = {"111": 961, "101": 15, "110": 14, "011": 9, "001": 1}
DIGITAL_CASES = [
    (("dephasing",), deph_res, 1),
    (("eff_noise",), deph_res, 1),
    (("depolarizing",), depo_res, 3),
    (("dephasing", "depolarizing"), deph_depo_res, 4),
    (("eff_noise", "dephasing"), eff_deph_res, 2),
    (("eff_noise", "leakage"), deph_res, 1),
    (("eff_noise", "leakage", "dephasing"), eff_deph_res, 2),
]


def seq_digital():
    reg = Register(
        {
            "control1": np.array([-4.0, 0.0]),
            "target": np.array([0.0, 4.0]),
            "control2": np.array([4.0, 0.0]),
        }
    )
    pi_y = Pulse.ConstantDetuning(BlackmanWaveform(1000, np.pi), 0.0, -np.pi / 2)
    seq = Sequence(reg, DigitalAnalogDevice)
    seq.declare_channel("raman", "raman_local", "control1")
    seq.add(pi_y, "raman")
    seq.target("target", "raman")
    seq.add(pi_y, "raman")
    seq.target("control2", "raman")
    seq.add(pi_y, "raman")
    return seq


def gen_noises_digital():
    for k, (noise, golden, n_ops) in enumerate(DIGITAL_CASES):
        np.random.seed(123)
        params = legacy_params(noise)
        if "dephasing" in noise:
            params["hyperfine_dephasing_rate"] = 0.05
        with_leakage = "leakage" in noise
        if with_leakage or "eff_noise" in noise:
            params["eff_noise_opers"] = [
                np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=complex)
                if with_leakage
                else Z2
            ]
            params["eff_noise_rates"] = [0.1 if with_leakage else 0.025]
        params.pop("runs", None)
        problems, aux, _ = capture(
            seq_digital(), NoiseModel(with_leakage=with_leakage, **params), 0.01
        )
        states = solve(problems[0], aux)
        counter = osamp.sample_state(
            states, aux["eval_times"], aux["eval_times"][-1], 1000,
            problems[0]["n_qudits"], problems[0]["eigenbasis"],
            aux["meas_basis"], aux["matching_meas_basis"],
        )
        ok = dict(counter) == golden
        print(f"noises_digital[{k}] {noise}: {'OK' if ok else 'MISMATCH'} {dict(counter)}")
        tight = solve(problems[0], aux, tight=True)
        idx = osamp.index_from_time(aux["eval_times"], aux["eval_times"][-1])
        P.save_problem(
            os.path.join(HERE, f"noises_digital_{k}.npz"),
            problems[0],
            aux=aux,
            seed=123,
            noise=list(noise),
            n_collapse_ops=n_ops,
            reference_golden_counter=golden,
            reference_cite="tests/pulser_simulation/test_simulation.py:1079-1160",
            oracle_final_state_default=states[-1],
            oracle_final_state_tight=tight[-1],
            oracle_lookup_index=idx,
            oracle_lookup_state_default=states[idx],
        )



# ---------------------------------------------------------------------------
# 2b. test_simulation.py:1179-1300 (3 atoms, "all" basis = digital + rydberg)
# ---------------------------------------------------------------------------

ALL_CASES = [
    (("dephasing",), {"111": 961, "101": 15, "110": 14, "011": 9, "001": 1}, 2),
    (("eff_noise",), {"111": 961, "101": 15, "110": 14, "011": 9, "001": 1}, 2),
    (("relaxation",), {"000": 459, "010": 202, "001": 168, "100": 167, "101": 4}, 1),
    (("dephasing", "relaxation"), {"000": 451, "010": 205, "001": 170, "100": 168, "101": 6}, 3),
    (("eff_noise", "dephasing"), {"111": 932, "101": 28, "011": 24, "110": 15, "001": 1}, 4),
    (("eff_noise", "leakage"), {"111": 961, "101": 15, "110": 14, "011": 9, "001": 1}, 2),
]


def seq_all():
    """The CCZ sequence fixture `seq` of test_simulation.py:76-96."""
    pi_pulse = Pulse.ConstantDetuning(BlackmanWaveform(1000, np.pi), 0.0, 0)
    twopi_pulse = Pulse.ConstantDetuning(BlackmanWaveform(1000, 2 * np.pi), 0.0, 0)
    seq = seq_digital()
    seq.declare_channel("ryd", "rydberg_local", "control1")
    seq.add(pi_pulse, "ryd", protocol="wait-for-all")
    seq.target("control2", "ryd")
    seq.add(pi_pulse, "ryd")
    seq.target("target", "ryd")
    seq.add(twopi_pulse, "ryd")
    seq.target("control2", "ryd")
    seq.add(pi_pulse, "ryd")
    seq.target("control1", "ryd")
    seq.add(pi_pulse, "ryd")
    seq.add(Pulse.ConstantPulse(1000, 1, 0, 0), "ryd")
    return seq


def gen_noises_all():
    pi_y = Pulse.ConstantDetuning(BlackmanWaveform(1000, np.pi), 0.0, -np.pi / 2)
    twopi_pulse = Pulse.ConstantDetuning(BlackmanWaveform(1000, 2 * np.pi), 0.0, 0)
    for k, (noise, golden, n_ops) in enumerate(ALL_CASES):
        seq = seq_all()
        params = {}
        if "relaxation" in noise:
            seq.target("control1", "raman")
            seq.add(pi_y, "raman")
            seq.target("target", "raman")
            seq.add(pi_y, "raman")
            seq.target("control2", "raman")
            seq.add(pi_y, "raman")
            seq.declare_channel("ryd_glob", "rydberg_global")
            seq.add(twopi_pulse, "ryd_glob")
            seq.measure()
            params["relaxation_rate"] = 1.0
        leak = "leakage" in noise
        dd = 4 if leak else 3
        deph_op = np.zeros((dd, dd), dtype=complex); deph_op[0, 0] = 1
        hyp_op = np.zeros((dd, dd), dtype=complex); hyp_op[2, 2] = 1
        if "dephasing" in noise:
            params["hyperfine_dephasing_rate"] = 0.1
            params["dephasing_rate"] = 0.1
        if leak or "eff_noise" in noise:
            params["eff_noise_opers"] = [deph_op, hyp_op]
            params["eff_noise_rates"] = [0.2, 0.2]
        problems, aux, _ = capture(seq, NoiseModel(with_leakage=leak, **params), 0.01)
        states = solve(problems[0], aux)
        np.random.seed(123)  # the reference seeds right before run()
        counter = osamp.sample_state(
            states, aux["eval_times"], aux["eval_times"][-1], 1000,
            problems[0]["n_qudits"], problems[0]["eigenbasis"],
            aux["meas_basis"], aux["matching_meas_basis"],
        )
        ok = dict(counter) == golden
        print(f"noises_all[{k}] {noise}: {'OK' if ok else 'MISMATCH'} {dict(counter)} meas={aux['meas_basis']}")
        tight = solve(problems[0], aux, tight=True)
        idx = osamp.index_from_time(aux["eval_times"], aux["eval_times"][-1])
        P.save_problem(
            os.path.join(HERE, f"noises_all_{k}.npz"), problems[0], aux=aux, seed=123,
            noise=list(noise), n_collapse_ops=n_ops, reference_golden_counter=golden,
            reference_cite="tests/pulser_simulation/test_simulation.py:1179-1300",
            oracle_final_state_default=states[-1], oracle_final_state_tight=tight[-1],
            oracle_lookup_index=idx, oracle_lookup_state_default=states[idx],
        )


# ---------------------------------------------------------------------------
# 2b'. test_simulation.py:889-918 (test_noise): CCZ sequence, SPAM with eta = 0.9,
#      15 trajectories x 5 samples, all-basis sesolve, seed 3
# ---------------------------------------------------------------------------
def gen_noise_spam_all():
    from collections import Counter
    from pulser_amd.pulser_adapter import sequence_inputs_from_pulser, problem_from_trajectory

    golden = {"000": 824, "100": 41, "101": 57, "001": 63, "010": 15}
    np.random.seed(3)
    seq = seq_all()
    nm = NoiseModel(samples_per_run=5, p_false_pos=0.01, p_false_neg=0.05, state_prep_error=0.9)
    samples = sampler.sample(seq, extended_duration=seq.get_duration())
    inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
    T = samples.max_duration
    ext = samples.extend_duration(T + 1)
    hd = HamiltonianData(ext, seq.register, seq.device, nm, 15)
    HamiltonianData(ext, seq.register, seq.device, NoiseModel(), 1)  # hidden noiseless draw
    rate = 0.01
    tlist = qp.sampling_times(T + 1, rate)
    eval_times = np.union1d(tlist, [0.0, T * 1e-3])
    opts = qp.default_options(channel_amp_det(ext), T)
    total = [Counter() for _ in eval_times]
    lookups, reps_list, bad_list = [], [], []
    idx = osamp.index_from_time(eval_times, eval_times[-1])
    for traj, noisy, reps in hd.noisy_samples:
        prob = problem_from_trajectory(hd, traj, noisy, reps, rate)
        ham = qp.build_hamiltonian(prob)
        psi0 = qp.all_ground_state(3, prob["eigenbasis"])
        states = qp.sesolve(ham, psi0, eval_times, **opts)
        for i, t in enumerate(eval_times):
            total[i] += osamp.sample_state(
                states, eval_times, t, 5 * reps, 3, prob["eigenbasis"], "digital", False,
                {"epsilon": 0.01, "epsilon_prime": 0.05})
        lookups.append(states[idx]); reps_list.append(reps); bad_list.append(prob["bad_atoms"])
    w = np.zeros(8)
    n_meas = sum(total[idx].values())
    for bs, c in total[idx].items():
        w[int(bs, 2)] = c / n_meas
    w = w / sum(w)
    final = osamp.get_samples(w, 1000, 3)
    ok = dict(final) == golden
    print(f"noise_spam_all: {'OK' if ok else 'MISMATCH'} {dict(final)} reps={reps_list}")
    P.save_problem(
        os.path.join(HERE, "noise_spam_all.npz"), {"inputs": inputs.to_dict()},
        seed=3, reference_golden_counter=golden,
        reference_cite="tests/pulser_simulation/test_simulation.py:889-923 (test_noise)",
        eval_times=eval_times, traj_reps=np.array(reps_list), traj_bad_atoms=np.array(bad_list),
        oracle_traj_lookup_states=np.stack(lookups),
        oracle_total_final_counter=dict(total[idx]),
    )


# ---------------------------------------------------------------------------
# 2b''. tests/pulser_simulation/test_simresults.py:63-90, 383-389, 446-456 (results_noisy):
#       2 atoms, global Blackman pi pulse, doppler + laser-waist amplitude + SPAM noise,
#       15 trajectories x 5 samples at all 1001 evaluation times, seed 123
# ---------------------------------------------------------------------------
def gen_results_noisy():
    from collections import Counter
    from pulser_amd.pulser_adapter import sequence_inputs_from_pulser, problem_from_trajectory

    golden = {"11": 676, "10": 295, "01": 137, "00": 126}
    reg = Register({"A": np.array([0.0, 0.0]), "B": np.array([0.0, 10.0])})
    seq = Sequence(reg, DigitalAnalogDevice)
    seq.declare_channel("ryd", "rydberg_global")
    seq.add(Pulse.ConstantDetuning(BlackmanWaveform(1000, np.pi), 0.0, 0), "ryd")
    params = dict(samples_per_run=5, temperature=50.0, state_prep_error=0.005, p_false_pos=0.01,
                  p_false_neg=0.05, amp_sigma=1e-3, laser_waist=175.0)
    nm = NoiseModel(**params)
    np.random.seed(123)
    samples = sampler.sample(seq, extended_duration=seq.get_duration())
    inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
    T = samples.max_duration
    ext = samples.extend_duration(T + 1)
    hd = HamiltonianData(ext, seq.register, seq.device, nm, 15)
    HamiltonianData(ext, seq.register, seq.device, NoiseModel(), 1)  # hidden noiseless draw
    tlist = qp.sampling_times(T + 1, 1.0)
    eval_times = np.union1d(tlist, [0.0, T * 1e-3])
    opts = qp.default_options(channel_amp_det(ext), T)
    total = [Counter() for _ in eval_times]
    all_states = []
    for traj, noisy, reps in hd.noisy_samples:
        prob = problem_from_trajectory(hd, traj, noisy, reps, 1.0)
        ham = qp.build_hamiltonian(prob)
        psi0 = qp.all_ground_state(2, prob["eigenbasis"])
        states = qp.sesolve(ham, psi0, eval_times, **opts)
        for i, t in enumerate(eval_times):
            total[i] += osamp.sample_state(
                states, eval_times, t, 5 * reps, 2, prob["eigenbasis"], "ground-rydberg", True,
                {"epsilon": 0.01, "epsilon_prime": 0.05})
        all_states.append(np.stack(states))
    idx = osamp.index_from_time(eval_times, eval_times[-1])
    n_meas = sum(total[idx].values())
    w = np.zeros(4)
    for bs, c in total[idx].items():
        w[int(bs, 2)] = c / n_meas
    expect_last = w[1] + w[3]  # <I x |r><r|>: atom B measured in 1 (test_simresults.py:388-389)
    w = w / sum(w)
    np.random.seed(123)
    final = osamp.get_samples(w, 1234, 2)
    ok = dict(final) == golden
    print(f"results_noisy: {'OK' if ok else 'MISMATCH'} {dict(final)}; expect[-1] = {expect_last:.4f} (reference: ~0.68)")
    P.save_problem(
        os.path.join(HERE, "results_noisy.npz"), {"inputs": inputs.to_dict()}, seed=123,
        noise_model=params, reference_golden_counter=golden, reference_expect_last=0.68,
        reference_cite="tests/pulser_simulation/test_simresults.py:63-90, 383-389, 446-456",
        eval_times=eval_times, oracle_traj_lookup_states=np.stack(all_states)[:, idx],
        oracle_total_final_counter=dict(total[idx]),
    )


# ---------------------------------------------------------------------------
# 2b-3. tests/pulser_simulation/test_simresults.py:244-275 (test_get_final_state_noisy):
#       raman_local pi pulse on atom A (digital basis), doppler + trap position
#       fluctuations + SPAM, 15 trajectories x 5 samples, seed 123
# ---------------------------------------------------------------------------
def gen_final_state_noisy():
    from collections import Counter
    from pulser_amd.pulser_adapter import sequence_inputs_from_pulser, problem_from_trajectory

    reg = Register({"A": np.array([0.0, 0.0]), "B": np.array([0.0, 10.0])})
    seq = Sequence(reg, DigitalAnalogDevice)
    seq.declare_channel("ram", "raman_local", initial_target="A")
    seq.add(Pulse.ConstantDetuning(BlackmanWaveform(1000, np.pi), 0.0, 0), "ram")
    params = dict(samples_per_run=5, temperature=50.0, trap_depth=0.01, trap_waist=0.02,
                  state_prep_error=0.005, p_false_pos=0.01, p_false_neg=0.05)
    nm = NoiseModel(**params)
    np.random.seed(123)
    samples = sampler.sample(seq, extended_duration=seq.get_duration())
    inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
    T = samples.max_duration
    ext = samples.extend_duration(T + 1)
    hd = HamiltonianData(ext, seq.register, seq.device, nm, 15)
    HamiltonianData(ext, seq.register, seq.device, NoiseModel(), 1)  # hidden noiseless draw
    tlist = qp.sampling_times(T + 1, 1.0)
    eval_times = np.union1d(tlist, [0.0, T * 1e-3])
    opts = qp.default_options(channel_amp_det(ext), T)
    total = [Counter() for _ in eval_times]
    lookup = []
    idx = osamp.index_from_time(eval_times, eval_times[-1])
    for traj, noisy, reps in hd.noisy_samples:
        prob = problem_from_trajectory(hd, traj, noisy, reps, 1.0)
        ham = qp.build_hamiltonian(prob)
        psi0 = qp.all_ground_state(2, prob["eigenbasis"])
        states = qp.sesolve(ham, psi0, eval_times, **opts)
        for i, t in enumerate(eval_times):
            total[i] += osamp.sample_state(
                states, eval_times, t, 5 * reps, 2, prob["eigenbasis"], "digital", True,
                {"epsilon": 0.01, "epsilon_prime": 0.05})
        lookup.append(states[idx])
    final = {k: v / 75 for k, v in total[-1].items()}
    golden = {"10": 0.96, "00": 0.04}
    print(f"final_state_noisy: {'OK' if final == golden else 'MISMATCH'} {final}")
    P.save_problem(
        os.path.join(HERE, "final_state_noisy.npz"), {"inputs": inputs.to_dict()}, seed=123,
        noise_model=params, reference_golden_results_last=golden,
        reference_cite="tests/pulser_simulation/test_simresults.py:244-275",
        eval_times=eval_times, oracle_traj_lookup_states=np.stack(lookup),
        traj_reps=np.array([r for _, _, r in hd.noisy_samples]),
    )


# ---------------------------------------------------------------------------
# 2b-4. tests/pulser_simulation/test_simulation.py:1928-2042 (effective size with
#       an SLM mask and SPAM bad atoms; mw / rydberg / raman global channels)
# ---------------------------------------------------------------------------
def gen_slm_effective_size():
    from pulser_amd.pulser_adapter import problem_from_trajectory, sequence_inputs_from_pulser

    reg = Register.square(2, prefix="atom")
    rise = Pulse.ConstantPulse(1500, 1, 0, 0)
    for ch in ("mw_global", "rydberg_global", "raman_global"):
        np.random.seed(15092021)
        seq = Sequence(reg, MockDevice)
        seq.declare_channel("ch0", ch)
        seq.add(rise, "ch0")
        seq.config_slm_mask(["atom1"])
        samples = sampler.sample(seq, extended_duration=seq.get_duration())
        inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
        T = samples.max_duration
        ext = samples.extend_duration(T + 1)
        params = dict(samples_per_run=5, state_prep_error=0.4, p_false_pos=0.01, p_false_neg=0.05)
        hd = HamiltonianData(ext, seq.register, seq.device, NoiseModel(**params), 15)
        traj, noisy, reps = next(iter(hd.noisy_samples))
        prob = problem_from_trajectory(hd, traj, noisy, reps, 0.01)
        bad = np.array([traj.bad_atoms[q] for q in reg.qubit_ids])
        assert list(bad) == [True, False, True, False]  # the reference's own assertion
        ham = qp.build_hamiltonian(prob)
        h0 = ham.matrix(0.0).toarray()
        extra = dict(seed=15092021, channel=ch, noise_model=params, traj0_bad_atoms=bad,
                     traj0_reps=reps, oracle_h0=h0, slm_end=int(samples._slm_mask.end),
                     reference_cite="tests/pulser_simulation/test_simulation.py:1928-2042")
        loc = prob["samples"]["Local"]
        for basis, per_atom in loc.items():
            for q, s in per_atom.items():
                for qty in ("amp", "det", "phase"):
                    extra[f"local__{basis}__{q}__{qty}"] = np.asarray(s[qty])
        print(f"slm_effective_size[{ch}]: bases {list(loc)}, |H(0)| = {np.abs(h0).sum():.4f}, dim {h0.shape}")
        P.save_problem(os.path.join(HERE, f"slm_effective_size_{ch}.npz"),
                       {"inputs": inputs.to_dict()}, **extra)


# ---------------------------------------------------------------------------
# 2b-5. tests/pulser_simulation/test_simulation.py:1748-1926 (SLM mask: XY masking
#       equals removing the qubit; masked first pulse; mask next to a local channel)
# ---------------------------------------------------------------------------
def gen_slm_masks():
    from pulser_amd.pulser_adapter import sequence_inputs_from_pulser

    def capture(seq):
        samples = sampler.sample(seq, extended_duration=seq.get_duration())
        return sequence_inputs_from_pulser(samples, seq.register, seq.device).to_dict()

    reg3 = Register({"q0": (0, 0), "q1": (10, 10), "q2": (-10, -10)})
    reg2 = Register({"q0": (0, 0), "q1": (10, 10)})
    pulse = Pulse.ConstantPulse(100, 10, 0, 0)
    no_pulse = Pulse.ConstantPulse(100, 0, 0, 0)
    out = {}
    # :1748-1789
    s = Sequence(reg3, MockDevice)
    s.set_magnetic_field(0, 1.0, 0.0)
    s.declare_channel("ch_masked", "mw_global")
    s.config_slm_mask(["q2"])
    s.add(pulse, "ch_masked")
    out["eq_masked"] = capture(s)
    s = Sequence(reg2, MockDevice)
    s.set_magnetic_field(0, 1.0, 0.0)
    s.declare_channel("ch_two", "mw_global")
    s.add(pulse, "ch_two")
    out["eq_two"] = capture(s)
    # :1792-1838
    s = Sequence(reg3, MockDevice)
    s.declare_channel("ch_masked", "mw_global")
    s.config_slm_mask(["q2"])
    for _ in range(3):
        s.add(pulse, "ch_masked")
    out["tp_masked"] = capture(s)
    mask_time = list(s._slm_mask_time)
    s = Sequence(reg3, MockDevice)
    s.declare_channel("ch_three", "mw_global")
    for p in (no_pulse, pulse, pulse):
        s.add(p, "ch_three")
    out["tp_three"] = capture(s)
    s = Sequence(reg2, MockDevice)
    s.declare_channel("ch_two", "mw_global")
    for p in (pulse, no_pulse, no_pulse):
        s.add(p, "ch_two")
    out["tp_two"] = capture(s)
    # :1841-1926
    s = Sequence(Register.square(2, prefix="q"), MockDevice)
    s.declare_channel("rydberg_global", "rydberg_global")
    s.config_slm_mask(["q0", "q3"])
    s.add(Pulse.ConstantPulse(1000, 10, 0, 0), "rydberg_global")
    s.declare_channel("raman_local", "raman_local", initial_target="q0")
    s.add(Pulse.ConstantPulse(1000, 10, -5, np.pi), "raman_local", protocol="no-delay")
    assert s._slm_mask_time == [0, 1000] and s._slm_mask_targets == {"q0", "q3"}
    out["local"] = capture(s)
    P.save_problem(os.path.join(HERE, "slm_masks.npz"), out, tp_mask_time=np.array(mask_time),
                   reference_cite="tests/pulser_simulation/test_simulation.py:1748-1926")
    print("slm_masks:", list(out), "mask time", mask_time)


# ---------------------------------------------------------------------------
# 2b-6. tests/pulser_simulation/test_simulation.py:2045-2153 (output modulation,
#       doppler + laser-waist amplitude noise, beam propagation direction)
# ---------------------------------------------------------------------------
def gen_modulation():
    import dataclasses

    from pulser.channels import Raman, Rydberg
    from pulser.devices import Device
    from pulser_amd.pulser_adapter import problem_from_trajectory, sequence_inputs_from_pulser

    reg = Register({"control1": np.array([-4.0, 0.0]), "target": np.array([0.0, 4.0]),
                    "control2": np.array([4.0, 0.0])})
    pulse1 = Pulse.ConstantPulse(120, 1, 0, 2.0)
    params = dict(samples_per_run=1, temperature=50.0, laser_waist=175.0)
    for tag, pdir in (("none", None), ("x", (1, 0, 0)), ("y", (0, 1, 0)), ("z", (0, 0, 1))):
        dev = Device(
            name="ModulatedDevice", dimensions=3, rydberg_level=70, max_atom_num=100,
            max_radial_distance=100, min_atom_distance=1,
            channel_objects=(
                Rydberg.Global(1000, 200, clock_period=1, min_duration=1, mod_bandwidth=4.0,
                               propagation_dir=pdir),
                Raman.Local(2 * np.pi * 20, 2 * np.pi * 10, max_targets=2, fixed_retarget_t=0,
                            min_retarget_interval=220, clock_period=4, mod_bandwidth=4.0),
            ),
        )
        seq = Sequence(reg, dev)
        seq.declare_channel("ch0", "rydberg_global")
        seq.declare_channel("ch1", "raman_local", initial_target="target")
        seq.add(pulse1, "ch1")
        seq.target("control1", "ch1")
        seq.add(pulse1, "ch1")
        seq.add(pulse1, "ch0")
        ch1 = seq.declared_channels["ch1"]
        mod = ch1.modulate(pulse1.amplitude.samples).as_array()
        mod_dt = pulse1.duration + pulse1.fall_time(ch1)
        np.random.seed(20260927)
        samples = sampler.sample(seq, modulation=True,
                                 extended_duration=seq.get_duration(include_fall_time=True))
        inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
        T = samples.max_duration
        ext = samples.extend_duration(T + 1)
        hd = HamiltonianData(ext, seq.register, seq.device, NoiseModel(**params), 15)
        traj, noisy, reps = next(iter(hd.noisy_samples))
        prob = problem_from_trajectory(hd, traj, noisy, reps, 1.0)
        extra = dict(seed=20260927, noise_model=params, mod_dt=int(mod_dt), modulated_pulse=mod,
                     propagation_dir=np.zeros(0) if pdir is None else np.array(pdir, float),
                     doppler=np.array([traj.doppler_detune[q] for q in reg.qubit_ids]),
                     reference_cite="tests/pulser_simulation/test_simulation.py:2045-2153")
        assert prob["samples"]["Global"] == {}
        for basis, per_atom in prob["samples"]["Local"].items():
            for q, s in per_atom.items():
                for qty in ("amp", "det", "phase"):
                    extra[f"local__{basis}__{q}__{qty}"] = np.asarray(s[qty])
        P.save_problem(os.path.join(HERE, f"modulation_dir_{tag}.npz"),
                       {"inputs": inputs.to_dict()}, **extra)
        print(f"modulation[{tag}]: duration {T}, mod_dt {mod_dt}")


# ---------------------------------------------------------------------------
# 2b-7. tests/pulser_simulation/test_simulation.py:2593-2660 (EOM mode at the
#       detuning limits, phase-drift correction; seeded final Counters)
# ---------------------------------------------------------------------------
def gen_eom_limit_det():
    from pulser.channels import Rydberg
    from pulser.channels.eom import RydbergBeam, RydbergEOM
    from pulser.devices import Device
    from pulser_amd.pulser_adapter import problem_from_trajectory, sequence_inputs_from_pulser

    goldens = {
        True: {"000": 850, "100": 53, "001": 46, "010": 42, "101": 9},
        False: {"000": 879, "010": 49, "100": 40, "001": 32},
    }
    reg = Register({"control1": np.array([-4.0, 0.0]), "target": np.array([0.0, 4.0]),
                    "control2": np.array([4.0, 0.0])})
    for min_det_on in (False, True):
        eom = RydbergEOM(mod_bandwidth=30.0, limiting_beam=RydbergBeam.RED,
                         max_limiting_amp=50 * 2 * np.pi, intermediate_detuning=800 * 2 * np.pi,
                         controlled_beams=(RydbergBeam.BLUE,) if min_det_on else (RydbergBeam.RED,))
        dev = Device(
            name="EomDevice", dimensions=3, rydberg_level=70, max_atom_num=2000,
            max_radial_distance=1000, min_atom_distance=1,
            channel_objects=(Rydberg.Global(1000, 200, clock_period=1, min_duration=1,
                                            mod_bandwidth=4.0, eom_config=eom),),
        )
        seq = Sequence(reg, dev)
        seq.declare_channel("ryd_glob", "rydberg_global")
        seq.add(Pulse.ConstantPulse(1000, np.pi / 2, 0, 0), "ryd_glob")
        max_abs_det = seq.declared_channels["ryd_glob"].max_abs_detuning
        det_on = -max_abs_det if min_det_on else max_abs_det
        seq.enable_eom_mode("ryd_glob", np.pi, det_on, correct_phase_drift=True)
        det_off = float(seq._schedule["ryd_glob"].eom_blocks[-1].detuning_off)
        det_on = float(det_on)
        assert det_off < det_on if min_det_on else det_off > det_on
        seq.add_eom_pulse("ryd_glob", 1000, 0)
        seq.delay(500, "ryd_glob")
        seq.modify_eom_setpoint("ryd_glob", np.pi / 2, 0, 0, correct_phase_drift=True)
        seq.add_eom_pulse("ryd_glob", 1000, 0)
        samples = sampler.sample(seq, extended_duration=seq.get_duration())
        inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
        T = samples.max_duration
        ext = samples.extend_duration(T + 1)
        np.random.seed(123)
        hd = HamiltonianData(ext, seq.register, seq.device, NoiseModel(), 1)
        traj, noisy, reps = next(iter(hd.noisy_samples))
        prob = problem_from_trajectory(hd, traj, noisy, reps, 1.0)
        ham = qp.build_hamiltonian(prob)
        tlist = qp.sampling_times(T + 1, 1.0)
        eval_times = np.union1d(tlist, [0.0, T * 1e-3])
        opts = qp.default_options(channel_amp_det(ext), T)
        psi0 = qp.all_ground_state(3, prob["eigenbasis"])
        states = qp.sesolve(ham, psi0, eval_times, **opts)
        tight = qp.sesolve(ham, psi0, np.array([0.0, eval_times[-1]]), **{**opts, **qp.TIGHT})[-1]
        idx = osamp.index_from_time(eval_times, eval_times[-1])
        final = osamp.sample_state(states, eval_times, eval_times[-1], 1000, 3, prob["eigenbasis"],
                                   "ground-rydberg", True, None)
        ok = dict(final) == goldens[min_det_on]
        print(f"eom_limit_det[min_detuning_on={min_det_on}]: {'OK' if ok else 'MISMATCH'} {dict(final)};"
              f" T = {T}, det_on {det_on:.2f}, det_off {det_off:.2f}")
        P.save_problem(
            os.path.join(HERE, f"eom_limit_det_{int(min_det_on)}.npz"), {"inputs": inputs.to_dict()},
            seed=123, reference_golden_counter=goldens[min_det_on], eval_times=eval_times,
            oracle_lookup_state=states[idx], oracle_final_state_tight=tight,
            detuning_on=det_on, detuning_off=det_off,
            reference_det=np.asarray(prob["samples"]["Global"]["ground-rydberg"]["det"]),
            reference_cite="tests/pulser_simulation/test_simulation.py:2593-2660",
        )


# ---------------------------------------------------------------------------
# 2b-8. Per-trajectory noisy samples of multi-channel sequences: test_simulation.py
#       :2193-2266 (amp_sigma per channel, two local channels sharing a basis),
#       :2422-2470 (the five test_noisy_runs noise models) and :1401-1427
#       (concurrent local + global pulses with doppler noise)
# ---------------------------------------------------------------------------
def gen_multichannel_noise():
    from pulser_amd.pulser_adapter import problem_from_trajectory, sequence_inputs_from_pulser

    def three_channel_seq(pulse, retarget):
        reg = Register({"q0": (0, 0), "q1": (10, 10)})
        seq = Sequence(reg, MockDevice)
        seq.declare_channel("ch0", "rydberg_global")
        seq.declare_channel("ch1", "raman_local", initial_target="q0")
        seq.declare_channel("ch2", "raman_local", initial_target="q1")
        seq.add(pulse, "ch0")
        seq.add(pulse, "ch0")
        seq.add(pulse, "ch1", protocol="no-delay")
        if retarget:
            seq.target("q1", "ch1")
            seq.add(pulse, "ch1", protocol="no-delay")
        seq.add(pulse, "ch2", protocol="no-delay")
        return seq

    def concurrent_seq():
        seq = Sequence(Register({"q0": (0, 0)}), DigitalAnalogDevice)
        seq.declare_channel("ch_local", "rydberg_local", initial_target="q0")
        seq.declare_channel("ch_global", "rydberg_global")
        pulse = Pulse.ConstantPulse(20, 10, 0, 0)
        seq.add(pulse, "ch_local")
        seq.add(pulse, "ch_global", protocol="no-delay")
        return seq

    zero = Pulse.ConstantPulse(10, 0, 0, 0)
    cases = {
        "amp_sigma": (three_channel_seq(Pulse.ConstantPulse(120, 1, 0, 2.0), True),
                      dict(amp_sigma=0.1), 3, 11),
        "runs_detuning_sigma": (three_channel_seq(zero, False), dict(detuning_sigma=1.0), 2, 1337),
        "runs_amp_sigma": (three_channel_seq(zero, False), dict(amp_sigma=1.0), 2, 1337),
        "runs_temperature": (three_channel_seq(zero, False), dict(temperature=10.0), 2, 1337),
        "runs_trap": (three_channel_seq(zero, False),
                      dict(temperature=10.0, disable_doppler=True, trap_depth=1000.0, trap_waist=0.1), 2, 1337),
        "runs_hf": (three_channel_seq(zero, False),
                    dict(detuning_hf_psd=(1.0, 2.0), detuning_hf_omegas=(3.0, 4.0)), 2, 1337),
        "concurrent": (concurrent_seq(), dict(samples_per_run=5, temperature=50.0), 15, 99),
    }
    out, extra = {}, {}
    for name, (seq, params, ntraj, seed) in cases.items():
        samples = sampler.sample(seq, extended_duration=seq.get_duration())
        out[name] = sequence_inputs_from_pulser(samples, seq.register, seq.device).to_dict()
        ext = samples.extend_duration(samples.max_duration + 1)
        np.random.seed(seed)
        hd = HamiltonianData(ext, seq.register, seq.device, NoiseModel(**params), ntraj)
        extra[f"{name}__rng_probe"] = np.random.get_state()[1][:4].copy()
        extra[f"{name}__noise_model"] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in params.items()}
        extra[f"{name}__seed"], extra[f"{name}__n_trajectories"] = seed, ntraj
        n_seen = 0
        for i, (traj, noisy, reps) in enumerate(hd.noisy_samples):
            prob = problem_from_trajectory(hd, traj, noisy, reps, 1.0)
            assert prob["samples"]["Global"] == {}
            for basis, per_atom in prob["samples"]["Local"].items():
                for q, s in per_atom.items():
                    for qty in ("amp", "det", "phase"):
                        extra[f"{name}__traj{i}__{basis}__{q}__{qty}"] = np.asarray(s[qty])
            extra[f"{name}__traj{i}__interaction"] = np.asarray(prob["interaction_matrix"])
            extra[f"{name}__traj{i}__reps"] = reps
            n_seen += 1
        extra[f"{name}__n_distinct"] = n_seen
        print(f"multichannel_noise[{name}]: {n_seen} distinct trajectories, "
              f"bases {list(prob['samples']['Local'])}")
    P.save_problem(os.path.join(HERE, "multichannel_noise.npz"), out,
                   reference_cite="tests/pulser_simulation/test_simulation.py:1401-1427, 2193-2266, 2422-2470",
                   **extra)


# ---------------------------------------------------------------------------
# 2c. test_simulation.py:1536-1690 (XY mode, SLM mask, SPAM trajectories, mesolve)
# ---------------------------------------------------------------------------

XY_CASES = [
    (None, "dephasing", {"0000": 830, "0001": 21, "0010": 3, "0100": 80, "1000": 66}, 1),
    (None, "eff_noise", {"0000": 851, "0001": 23, "0010": 8, "0100": 57, "1000": 61}, 1),
    (None, "leakage", {"0000": 851, "0001": 23, "0010": 8, "0100": 57, "1000": 61}, 1),
    (None, "depolarizing", {"0000": 791, "0001": 39, "0010": 10, "0100": 81, "0110": 2, "1000": 67, "1010": 10}, 3),
    ("atom0", "dephasing", {"0000": 804, "0001": 105, "0010": 12, "0100": 54, "0101": 8, "1000": 17}, 1),
    ("atom1", "dephasing", {"0000": 575, "0001": 334, "0011": 12, "0100": 13, "1000": 56, "1001": 10}, 1),
]


def gen_noisy_xy():
    from collections import Counter
    from pulser_amd.pulser_adapter import sequence_inputs_from_pulser, problem_from_trajectory

    seed = 15092021
    for k, (masked, noise, golden, n_ops) in enumerate(XY_CASES):
        np.random.seed(seed)
        reg = Register.square(2, prefix="atom")
        seq = Sequence(reg, MockDevice)
        seq.declare_channel("ch0", "mw_global")
        if masked is not None:
            seq.config_slm_mask([masked])
        seq.add(Pulse.ConstantPulse(1000, 3.0, 1.0, 0.0), "ch0")
        leak = noise == "leakage"
        if leak or noise == "eff_noise":
            op = np.diag([1.0, -1.0, 0.0]).astype(complex) if leak else np.diag([1.0, -1.0]).astype(complex)
            params = dict(eff_noise_opers=[op], eff_noise_rates=[1.0])
        else:
            params = {f"{noise}_rate": _LEGACY_DEFAULTS[f"{noise}_rate"]}
        nm = NoiseModel(samples_per_run=10, with_leakage=leak, state_prep_error=0.4,
                        p_false_pos=0.01, p_false_neg=0.05, **params)
        samples = sampler.sample(seq, extended_duration=seq.get_duration())
        inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
        T = samples.max_duration
        ext = samples.extend_duration(T + 1)
        hd = HamiltonianData(ext, seq.register, seq.device, nm, 15)
        HamiltonianData(ext, seq.register, seq.device, NoiseModel(), 1)  # hidden noiseless draw
        rate = 0.1
        tlist = qp.sampling_times(T + 1, rate)
        eval_times = np.union1d(tlist, [0.0, T * 1e-3])
        opts = qp.default_options(channel_amp_det(ext), T)
        total = [Counter() for _ in eval_times]
        finals, reps_list, bad_list = [], [], []
        for traj, noisy, reps in hd.noisy_samples:
            prob = problem_from_trajectory(hd, traj, noisy, reps, rate)
            ham = qp.build_hamiltonian(prob)
            psi0 = qp.all_ground_state(4, prob["eigenbasis"], xy=True)
            states = qp.mesolve(ham, psi0, eval_times, **opts)
            for i, t in enumerate(eval_times):
                total[i] += osamp.sample_state(
                    states, eval_times, t, 10 * reps, 4, prob["eigenbasis"], "XY", True,
                    {"epsilon": 0.01, "epsilon_prime": 0.05})
            idx = osamp.index_from_time(eval_times, eval_times[-1])
            finals.append(states[idx]); reps_list.append(reps)
            bad_list.append(prob["bad_atoms"])
        # NoisyResults.sample_final_state: resample the final SampledResult (result.py:171-242)
        idx = osamp.index_from_time(eval_times, eval_times[-1])
        w = np.zeros(16)
        n_meas = sum(total[idx].values())
        for bs, c in total[idx].items():
            w[int(bs, 2)] = c / n_meas
        w = w / sum(w)
        final = osamp.get_samples(w, 1000, 4)
        ok = dict(final) == golden
        print(f"noisy_xy[{k}] masked={masked} {noise}: {'OK' if ok else 'MISMATCH'} {dict(final)}")
        P.save_problem(
            os.path.join(HERE, f"noisy_xy_{k}.npz"), {"inputs": inputs.to_dict()},
            seed=seed, masked=masked or "", noise=noise, n_collapse_ops=n_ops,
            reference_golden_counter=golden,
            reference_cite="tests/pulser_simulation/test_simulation.py:1536-1690 (MESOLVER cases)",
            eval_times=eval_times, traj_reps=np.array(reps_list), traj_bad_atoms=np.array(bad_list),
            oracle_traj_lookup_states=np.stack(finals),
            oracle_total_final_counter=dict(total[idx]),
        )

# ---------------------------------------------------------------------------
# 3. test_simulation.py:2156-2190 (3 atoms, custom initial state, golden state)
# ---------------------------------------------------------------------------

GOLDEN_3ATOM = np.array(
    [
        0.28985369 + 0.13530479j,
        0.40220557 + 0.0j,
        0.27445983 + 0.15541026j,
        0.29608403 + 0.06155379j,
        0.40220557 + 0.0j,
        0.36173532 - 0.01617572j,
        0.29608403 + 0.06155379j,
        0.36931122 - 0.15570528j,
    ]
)


def gen_three_atom_state():
    seq = Sequence(
        Register({"q0": (-6, 0), "q1": (0, 0), "q2": (6, 0)}), AnalogDevice
    )
    seq.declare_channel("ising", "rydberg_global")
    seq.add(Pulse.ConstantPulse(4000, 9.28, 18.7, 0), "ising")
    problems, aux, _ = capture(seq, None)
    psi0 = np.ones(8, dtype=complex) / np.sqrt(8)  # .unit() simulation.py:523-525
    states = solve(problems[0], aux, psi0=psi0)
    tight = solve(problems[0], aux, psi0=psi0, tight=True)
    idx = osamp.index_from_time(aux["eval_times"], aux["eval_times"][-1])
    st = states[idx]
    ph = np.angle(st[np.argmax(np.abs(st))])
    err = np.max(np.abs(st * np.exp(-1j * ph) - GOLDEN_3ATOM))
    print(f"three_atom_state: lookup idx {idx} of {len(states)}; |oracle - golden| = {err:.2e}")
    P.save_problem(
        os.path.join(HERE, "three_atom_state.npz"),
        problems[0],
        aux=aux,
        initial_state=psi0,
        reference_golden_state=GOLDEN_3ATOM,
        reference_cite="tests/pulser_simulation/test_simulation.py:2156-2190 (rtol 1e-2, global phase removed)",
        oracle_lookup_index=idx,
        oracle_states_default=np.stack([states[0], states[len(states) // 2], states[idx], states[-1]]),
        oracle_states_tight=np.stack([tight[0], tight[len(tight) // 2], tight[idx], tight[-1]]),
        oracle_state_indices=np.array([0, len(states) // 2, idx, len(states) - 1]),
    )


# ---------------------------------------------------------------------------
# 4. BASELINE.json configs
# ---------------------------------------------------------------------------


def anneal_sequence(reg, device=MockDevice):
    """tests/pulser_simulation/test_qutip_backend_v2.py:56-88 on ``reg``."""
    omega_max = 4 * 2 * np.pi
    U = omega_max / 2
    d0, df = -6 * U, 2 * U
    t_rise, t_fall = 500, 1000
    t_sweep = int((df - d0) / (2 * np.pi * 10) * 1000)
    seq = Sequence(reg, device)
    seq.declare_channel("ising_global", "rydberg_global")
    seq.add(Pulse.ConstantDetuning(RampWaveform(t_rise, 0.0, omega_max), d0, 0.0), "ising_global")
    seq.add(Pulse.ConstantAmplitude(omega_max, RampWaveform(t_sweep, d0, df), 0.0), "ising_global")
    seq.add(Pulse.ConstantDetuning(RampWaveform(t_fall, omega_max, 0.0), df, 0.0), "ising_global")
    return seq


def blockade_radius():
    return MockDevice.rydberg_blockade_radius(4 * 2 * np.pi / 2)


def pick(states, idxs):
    return np.stack([states[i] for i in idxs])


def gen_cfg1():
    """4-atom square, global Blackman pi pulse, sesolve (plumbing)."""
    reg = Register.square(2, spacing=5.0, prefix="q")
    seq = Sequence(reg, MockDevice)
    seq.declare_channel("ryd", "rydberg_global")
    seq.add(Pulse.ConstantDetuning(BlackmanWaveform(1000, np.pi), 0.0, 0.0), "ryd")
    problems, aux, _ = capture(seq, None)
    p = problems[0]
    # the synthetic generators must reproduce pulser's inputs exactly
    coords = P.register_coords(P.square_rect(2, 2), 5.0)
    assert np.array_equal(coords, p["coords"]), (coords, p["coords"])
    syn_amp = np.concatenate([P.blackman_samples(1000, np.pi), [0.0]])
    assert np.array_equal(syn_amp, p["samples"]["Global"]["ground-rydberg"]["amp"])
    assert np.array_equal(P.interaction_matrix(coords, P.C6_LEVEL70), p["interaction_matrix"])
    states = solve(p, aux)
    tight = solve(p, aux, tight=True)
    idx = osamp.index_from_time(aux["eval_times"], aux["eval_times"][-1])
    np.random.seed(123)
    counter = osamp.sample_state(states, aux["eval_times"], aux["eval_times"][-1], 1000, 4, p["eigenbasis"], "ground-rydberg")
    sel = [0, 250, 500, 750, idx, len(states) - 1]
    print(f"cfg1: lookup idx {idx}; counter {dict(counter)}")
    P.save_problem(
        os.path.join(HERE, "cfg1_square4_pi.npz"), p, aux=aux, seed=123,
        oracle_counter_default=dict(counter),
        oracle_lookup_index=idx,
        oracle_state_indices=np.array(sel),
        oracle_states_default=pick(states, sel),
        oracle_states_tight=pick(tight, sel),
    )


def gen_cfg2(n=12, name="cfg2_chain12_anneal"):
    """n-atom chain at the blockade radius, analog Ising anneal, sesolve."""
    rb = blockade_radius()
    reg = Register.rectangle(1, n, rb, prefix="q")
    problems, aux, _ = capture(anneal_sequence(reg), None)
    p = problems[0]
    coords = P.register_coords(P.square_rect(1, n), rb)
    assert np.allclose(coords, p["coords"], rtol=0, atol=1e-12)
    syn = P.anneal_samples()
    for k in ("amp", "det", "phase"):
        assert np.array_equal(syn[k], p["samples"]["Global"]["ground-rydberg"][k]), k
    assert np.allclose(P.interaction_matrix(p["coords"], P.C6_LEVEL70), p["interaction_matrix"], rtol=1e-15)
    aux_min = dict(aux)
    sel_t = np.array([0.0, 0.5, 1.3, 2.1, 3.099, 3.1])
    aux_min["eval_times"] = sel_t
    counter = [0]
    import time
    t0 = time.time()
    states = qp.sesolve(qp.build_hamiltonian(p), qp.all_ground_state(n, p["eigenbasis"]), sel_t, counter=counter, **aux["options"])
    wall = time.time() - t0
    tight = solve(p, aux_min, tight=True)
    print(f"{name}: default zvode {counter[0]} RHS evals in {wall:.2f}s; |default-tight|max = {np.max(np.abs(states[-1]-tight[-1])):.2e}; norm drift {np.linalg.norm(states[-1])-1:.2e}")
    # keep the fixture small: only the scalar inputs + the states
    small = {k: v for k, v in p.items() if k != "samples"}
    small["samples"] = {"Global": {}, "Local": {}}  # regenerated by pulser_amd.problem.anneal_samples
    P.save_problem(
        os.path.join(HERE, name + ".npz"), small, aux={k: v for k, v in aux.items() if k not in ("eval_times", "channel_amp_det")},
        synthetic="anneal_samples() on a 1 x n chain at the blockade radius",
        blockade_radius=float(rb),
        eval_times=sel_t,
        oracle_states_default=np.stack(states),
        oracle_states_tight=np.stack(tight),
        oracle_rhs_evals_default=counter[0],
    )


def gen_cfg3_small(rows=2, cols=3):
    """cfg3 physics (triangular register, dephasing + SPAM measurement errors,
    mesolve) at an oracle-sized N."""
    rb = blockade_radius()
    reg = Register.triangular_lattice(rows, cols, rb, prefix="q")
    nm = NoiseModel(dephasing_rate=0.05, p_false_pos=0.01, p_false_neg=0.05)
    np.random.seed(7)
    problems, aux, _ = capture(anneal_sequence(reg), nm)
    p = problems[0]
    n = p["n_qudits"]
    coords = P.register_coords(P.triangular_rect(rows, cols), rb)
    assert np.allclose(coords, p["coords"], rtol=0, atol=1e-12), (coords, p["coords"])
    sel_t = np.array([0.0, 0.5, 1.3, 2.1, 3.099, 3.1])
    aux2 = dict(aux)
    aux2["eval_times"] = sel_t
    states = solve(p, aux2)
    tight = solve(p, aux2, tight=True)
    np.random.seed(123)
    w = osamp.weights(states[-2], n, p["eigenbasis"], "ground-rydberg")
    c0 = osamp.get_samples(w, 1000, n)
    c1 = osamp.spam_flips(c0, 0.01, 0.05)
    print(f"cfg3_small N={n}: collapse {p['collapse_ops']}; trace {np.trace(states[-1]).real:.8f}; |default-tight| {np.max(np.abs(states[-1]-tight[-1])):.2e}")
    small = {k: v for k, v in p.items() if k != "samples"}
    small["samples"] = {"Global": {}, "Local": {}}
    P.save_problem(
        os.path.join(HERE, f"cfg3_tri{n}_dephasing.npz"), small,
        aux={k: v for k, v in aux.items() if k not in ("eval_times", "channel_amp_det")},
        rows=rows, cols=cols, blockade_radius=float(rb),
        eval_times=sel_t, seed=123,
        meas_errors={"epsilon": 0.01, "epsilon_prime": 0.05},
        oracle_states_default=np.stack(states),
        oracle_states_tight=np.stack(tight),
        oracle_counter_t3099_noflip=dict(c0),
        oracle_counter_t3099_spam=dict(c1),
    )



def gen_ns_tri14(rows=2, cols=7, name="ns_tri14_anneal"):
    """The bench headline at FULL size (SURVEY 8c last row): 14-atom triangular register (2 x 7,
    spacing R_b), the anneal of test_qutip_backend_v2.py:56-88, sesolve; tight oracle only
    (zvode rtol 1e-13) at six times incl. T.  Minutes of one core."""
    import time
    rb = blockade_radius()
    reg = Register.triangular_lattice(rows, cols, rb, prefix="q")
    problems, aux, _ = capture(anneal_sequence(reg), None)
    p = problems[0]
    n = p["n_qudits"]
    coords = P.register_coords(P.triangular_rect(rows, cols), rb)
    assert np.allclose(coords, p["coords"], rtol=0, atol=1e-12)
    syn = P.anneal_samples()
    for k in ("amp", "det", "phase"):
        assert np.array_equal(syn[k], p["samples"]["Global"]["ground-rydberg"][k]), k
    sel_t = np.array([0.0, 0.5, 1.3, 2.1, 3.099, 3.1])
    aux_min = dict(aux)
    aux_min["eval_times"] = sel_t
    counter = [0]
    t0 = time.time()
    opts = dict(aux["options"])
    opts.update(qp.TIGHT)
    tight = qp.sesolve(qp.build_hamiltonian(p), qp.all_ground_state(n, p["eigenbasis"]), sel_t, counter=counter, **opts)
    print(f"{name}: tight zvode {counter[0]} RHS evals in {time.time() - t0:.1f}s; norm drift {np.linalg.norm(tight[-1]) - 1:.2e}", flush=True)
    small = {k: v for k, v in p.items() if k != "samples"}
    small["samples"] = {"Global": {}, "Local": {}}
    P.save_problem(
        os.path.join(HERE, name + ".npz"), small, aux={k: v for k, v in aux.items() if k not in ("eval_times", "channel_amp_det")},
        synthetic=f"anneal_samples() on a {rows} x {cols} triangular register at the blockade radius",
        rows=rows, cols=cols, blockade_radius=float(rb),
        eval_times=sel_t,
        oracle_states_tight=np.stack(tight),
        oracle_rhs_evals_tight=counter[0],
    )


def gen_ns_rect16(rows=4, cols=4, name="ns_rect16_anneal"):
    """SURVEY 8(d) cfg5 row, "oracle at N <= 16": 16-atom 4 x 4 square register at the blockade radius, the same anneal,
    sesolve, tight oracle (zvode rtol 1e-13) at four times incl. T - the register size from which the split-operator
    passes (and the Lanczos alternative) are the product's default.  Tens of minutes of one core; 4 MiB of kets."""
    import time
    rb = blockade_radius()
    reg = Register.rectangle(rows, cols, rb, prefix="q")
    problems, aux, _ = capture(anneal_sequence(reg), None)
    p = problems[0]
    n = p["n_qudits"]
    coords = P.register_coords(P.square_rect(rows, cols), rb)
    assert np.allclose(coords, p["coords"], rtol=0, atol=1e-12), (coords, p["coords"])
    syn = P.anneal_samples()
    for k in ("amp", "det", "phase"):
        assert np.array_equal(syn[k], p["samples"]["Global"]["ground-rydberg"][k]), k
    sel_t = np.array([0.0, 0.5, 1.3, 2.1, 3.1])
    counter = [0]
    t0 = time.time()
    opts = dict(aux["options"])
    opts.update(qp.TIGHT)
    tight = qp.sesolve(qp.build_hamiltonian(p), qp.all_ground_state(n, p["eigenbasis"]), sel_t, counter=counter, **opts)
    print(f"{name}: tight zvode {counter[0]} RHS evals in {time.time() - t0:.1f}s; norm drift {np.linalg.norm(tight[-1]) - 1:.2e}", flush=True)
    small = {k: v for k, v in p.items() if k != "samples"}
    small["samples"] = {"Global": {}, "Local": {}}
    P.save_problem(
        os.path.join(HERE, name + ".npz"), small, aux={k: v for k, v in aux.items() if k not in ("eval_times", "channel_amp_det")},
        synthetic=f"anneal_samples() on a {rows} x {cols} square register at the blockade radius",
        rows=rows, cols=cols, blockade_radius=float(rb),
        eval_times=sel_t,
        oracle_states_tight=np.stack(tight[1:]),  # (t = 0 is the all-ground ket)
        oracle_rhs_evals_tight=counter[0],
    )


def gen_cfg3_tight(rows=2, cols=4, t_stop=None):
    """cfg3 physics on an INTERACTING triangular register large enough for the split-operator row
    path (k_ket rows need >= 10 atoms; 8 atoms for the multi-launch kernels): tight oracle only."""
    import time
    rb = blockade_radius()
    reg = Register.triangular_lattice(rows, cols, rb, prefix="q")
    nm = NoiseModel(dephasing_rate=0.05, p_false_pos=0.01, p_false_neg=0.05)
    np.random.seed(7)
    problems, aux, _ = capture(anneal_sequence(reg), nm)
    p = problems[0]
    n = p["n_qudits"]
    coords = P.register_coords(P.triangular_rect(rows, cols), rb)
    assert np.allclose(coords, p["coords"], rtol=0, atol=1e-12), (coords, p["coords"])
    sel_t = np.array([0.0, 0.5, 1.3, 2.1, 3.099, 3.1])
    if t_stop is not None:  # (cfg3_12_to1300: the first 1.3 us only - the 12-atom right-hand side takes ~1 s, the whole anneal hours)
        sel_t = sel_t[sel_t <= t_stop + 1e-12]
    opts = dict(aux["options"])
    opts.update(qp.TIGHT)
    counter = [0]
    t0 = time.time()
    ham = qp.build_hamiltonian(p)
    if n >= 9:
        from oracle import fast_lindblad as fl  # C restatement of lindblad_rhs, checked against it below
        rhs = fl.lindblad_rhs_fast(p, ham)
        y = np.random.default_rng(5).standard_normal(2 * 4**n).view(complex)
        a, b = rhs(1.234, y), qp.lindblad_rhs(ham)(1.234, y)
        print(f"fast rhs vs scipy rhs: {np.max(np.abs(a - b)):.2e}", flush=True)
        assert np.max(np.abs(a - b)) < 1e-11
        psi0 = qp.all_ground_state(n, p["eigenbasis"])
        ys = qp._zvode(rhs, np.outer(psi0, psi0.conj()).ravel(), sel_t, opts, counter)
        tight = [y.reshape(2**n, 2**n) for y in ys]
    else:
        tight = qp.mesolve(ham, qp.all_ground_state(n, p["eigenbasis"]), sel_t, counter=counter, **opts)
    print(f"cfg3_tight N={n}: {counter[0]} RHS evals in {time.time() - t0:.1f}s; trace {np.trace(tight[-1]).real:.12f}", flush=True)
    small = {k: v for k, v in p.items() if k != "samples"}
    small["samples"] = {"Global": {}, "Local": {}}
    P.save_problem(
        os.path.join(HERE, f"cfg3_tri{n}_dephasing" + ("" if t_stop is None else f"_to{round(t_stop * 1000)}") + ".npz"), small,
        aux={k: v for k, v in aux.items() if k not in ("eval_times", "channel_amp_det")},
        rows=rows, cols=cols, blockade_radius=float(rb),
        eval_times=sel_t, seed=123,
        meas_errors={"epsilon": 0.01, "epsilon_prime": 0.05},
        oracle_rhs_evals_tight=counter[0],
        **sketch_density_matrices(tight, n),
    )


SKETCH_ROWS, SKETCH_PROBES, SKETCH_SEED = 32, 4, 11


def sketch_rows(D):
    """Rows kept in full: first, last (all-ground), and a seeded draw."""
    rng = np.random.default_rng(SKETCH_SEED)
    return np.unique(np.concatenate([[0, D - 1], rng.choice(D, min(D, SKETCH_ROWS) - 2, replace=False)]))


def sketch_probes(D):
    rng = np.random.default_rng(SKETCH_SEED + 1)
    return rng.standard_normal((D, SKETCH_PROBES)) + 1j * rng.standard_normal((D, SKETCH_PROBES))


def sketch_density_matrices(tight, n):
    """A 10-atom rho is 16 MiB per time - too much for a fixture.  Kept instead: 32 full rows, the diagonal,
    rho @ V for 4 seeded Gaussian probe vectors (an error of size eps ANYWHERE in rho shows up as ~eps in
    the product of its row), trace and purity; up to 8 atoms the full upper triangle too."""
    D = 2**n
    rows, V = sketch_rows(D), sketch_probes(D)
    keep = 8 if n >= 12 else 0  # (12 atoms: every 4th of the 32 drawn rows - 32 rows x 6 times would be 12.6 MB)
    if keep:
        rows = rows[np.round(np.linspace(0, len(rows) - 1, keep)).astype(int)]
    out = {
        "oracle_rows": rows,
        "oracle_rows_tight": np.stack([t[rows] for t in tight]),
        "oracle_diag_tight": np.stack([np.diag(t) for t in tight]),
        "oracle_probe_products_tight": np.stack([t @ V for t in tight]),
        "oracle_purity_tight": np.array([np.vdot(t, t).real for t in tight]),
        "oracle_hermiticity_defect": max(float(np.max(np.abs(t - t.conj().T))) for t in tight),
        "sketch": {"rows": SKETCH_ROWS, "probes": SKETCH_PROBES, "seed": SKETCH_SEED, **({"keep": keep} if keep else {})},
    }
    if n <= 8:
        iu = np.triu_indices(D)
        out["oracle_states_tight_triu"] = np.stack([t[iu] for t in tight])
    return out


def thin_sketch(name, keep=8):
    """Rewrite a sketch fixture that was generated with all 32 rows (cfg3_tri12_dephasing.npz: 7.8 h of CPU before the
    `keep` rule of sketch_density_matrices existed) with the rows that rule keeps - the file a fresh run would write."""
    path = os.path.join(HERE, name)
    problem, extra = P.load_problem(path)
    rows = np.asarray(extra["oracle_rows"])
    sel = np.round(np.linspace(0, len(rows) - 1, keep)).astype(int)
    extra["oracle_rows"] = rows[sel]
    extra["oracle_rows_tight"] = np.asarray(extra["oracle_rows_tight"])[:, sel]
    extra["sketch"] = dict(extra["sketch"], keep=keep)
    P.save_problem(path, problem, **extra)


def gen_cfg4(n=12, ntraj=1024, keep=3):
    """12-atom chain, doppler + amplitude + SPAM noise, seed 0, 1024 trajectories."""
    rb = blockade_radius()
    reg = Register.rectangle(1, n, rb, prefix="q")
    nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.005,
                    p_false_pos=0.01, p_false_neg=0.05)
    np.random.seed(0)
    samples = sampler.sample(anneal_sequence(reg), extended_duration=3100)
    ext = samples.extend_duration(3101)
    hd = HamiltonianData(ext, reg, MockDevice, nm, ntraj)
    rng_after = np.random.get_state()[1][:4].copy()
    qids = list(reg.qubits)
    bad = np.array([[t.trajectory.bad_atoms[q] for q in qids] for t in hd.noise_trajectories])
    dop = np.array([[t.trajectory.doppler_detune[q] for q in qids] for t in hd.noise_trajectories])
    ampf = np.array([t.trajectory.amp_fluctuations["ising_global"] for t in hd.noise_trajectories])
    reps = np.array([t.reps for t in hd.noise_trajectories])
    it = hd.noisy_samples
    from pulser_amd.pulser_adapter import problem_from_trajectory
    kept = []
    for _ in range(keep):
        traj, noisy, r = next(it)
        kept.append(problem_from_trajectory(hd, traj, noisy, r, 1.0))
    sel_t = np.array([0.0, 1.3, 3.099, 3.1])
    opts = qp.default_options(channel_amp_det(ext), 3100)
    outs, outs_t = [], []
    for p in kept:
        ham = qp.build_hamiltonian(p)
        psi0 = qp.all_ground_state(n, p["eigenbasis"])
        outs.append(np.stack(qp.sesolve(ham, psi0, sel_t, **opts)))
        o2 = dict(opts); o2.update(qp.TIGHT)
        outs_t.append(np.stack(qp.sesolve(ham, psi0, sel_t, **o2)))
    print(f"cfg4: {len(hd.noise_trajectories)} trajectories, bad atoms total {bad.sum()}, amp mean {ampf.mean():.4f}; kept {keep}")
    # per-trajectory samples are affine in the base samples -> store only traj 0 in full
    first = kept[0]
    P.save_problem(
        os.path.join(HERE, "cfg4_chain12_noise.npz"), first,
        seed=0, n_trajectories=ntraj,
        noise_model=dict(temperature=50.0, amp_sigma=0.05, state_prep_error=0.005, p_false_pos=0.01, p_false_neg=0.05),
        blockade_radius=float(rb),
        traj_bad_atoms=bad, traj_doppler=dop, traj_amp_fluctuation=ampf, traj_reps=reps,
        rng_probe_after_ctor=rng_after,
        options={k: float(v) for k, v in opts.items()},
        eval_times=sel_t,
        kept_det=np.stack([[p["samples"]["Local"]["ground-rydberg"][q]["det"] for q in range(n)] for p in kept])[:, :, ::100],
        kept_amp=np.stack([[p["samples"]["Local"]["ground-rydberg"][q]["amp"] for q in range(n)] for p in kept])[:, :, ::100],
        kept_interaction=np.stack([p["interaction_matrix"] for p in kept]),
        oracle_states_default=np.stack(outs),
        oracle_states_tight=np.stack(outs_t),
    )


def gen_dmm():
    """Detuning-map-modulator channel next to a global Rydberg channel (4 atoms):
    noiseless, and with dmm_sigma + crosstalk (spot waist) + doppler noise.  The
    per-trajectory nested samples come from pulser-core's HamiltonianData."""
    from pulser_amd.pulser_adapter import problem_from_trajectory, sequence_inputs_from_pulser

    reg = Register.rectangle(2, 2, 6.0, prefix="q")
    det_map = reg.define_detuning_map({"q0": 0.1, "q1": 0.4, "q2": 0.2, "q3": 0.3})
    seq = Sequence(reg, MockDevice)
    seq.config_detuning_map(det_map, "dmm_0")
    seq.declare_channel("ising", "rydberg_global")
    seq.add(Pulse.ConstantPulse(200, 2 * np.pi, -3.0, 0.2), "ising")
    seq.add_dmm_detuning(RampWaveform(200, -20.0, -4.0), "dmm_0")
    seq.add(Pulse.ConstantAmplitude(5.0, RampWaveform(152, -6.0, 8.0), 0.0), "ising", protocol="no-delay")
    T = seq.get_duration()
    samples = sampler.sample(seq, extended_duration=T)
    inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
    ext = samples.extend_duration(T + 1)
    sel_t = np.array([0.0, 0.2, T * 1e-3])
    cases = {
        "noiseless": (NoiseModel(), 1),
        "noisy": (NoiseModel(dmm_sigma=0.2, detuning_map_spot_waist=4.0, temperature=40.0,
                             state_prep_error=0.1, runs=5, samples_per_run=1), 5),
    }
    extra = {}
    for name, (nm, ntraj) in cases.items():
        np.random.seed(77)
        hd = HamiltonianData(ext, seq.register, seq.device, nm, ntraj)
        probe = np.random.get_state()[1][:4].copy()
        opts = qp.default_options(channel_amp_det(ext), T)
        opts.update(qp.TIGHT)
        dets, amps, finals, dmmf, bad = [], [], [], [], []
        for traj, noisy, reps in hd.noisy_samples:
            prob = problem_from_trajectory(hd, traj, noisy, reps, 1.0)
            nested = prob["samples"]
            if nested["Local"].get("ground-rydberg"):
                loc = nested["Local"]["ground-rydberg"]
                dets.append(np.stack([loc[q]["det"] for q in range(4)]))
                amps.append(np.stack([loc[q]["amp"] for q in range(4)]))
            else:
                g = nested["Global"]["ground-rydberg"]
                dets.append(np.stack([g["det"]] * 4))
                amps.append(np.stack([g["amp"]] * 4))
            ham = qp.build_hamiltonian(prob)
            psi0 = qp.all_ground_state(4, prob["eigenbasis"])
            finals.append(np.stack(qp.sesolve(ham, psi0, sel_t, **opts)))
            dmmf.append(traj.dmm_det_fluctuation["dmm_0"])
            bad.append([traj.bad_atoms[q] for q in reg.qubits])
        extra[name] = dict(det=np.stack(dets), amp=np.stack(amps), states=np.stack(finals),
                           dmm_factor=np.array(dmmf), bad=np.array(bad), rng_probe=probe)
        print(f"dmm[{name}]: {len(dets)} trajectories, dmm factors {dmmf}")
    P.save_problem(
        os.path.join(HERE, "dmm_square4.npz"), {"inputs": inputs.to_dict()},
        seed=77, eval_times=sel_t,
        noisy_model=dict(dmm_sigma=0.2, detuning_map_spot_waist=4.0, temperature=40.0,
                         state_prep_error=0.1, runs=5, samples_per_run=1),
        **{f"{k}_{q}": v for k, d in extra.items() for q, v in d.items()},
    )


def gen_waist():
    """Amplitude noise with a finite laser waist (hamiltonian_data.py:758-780,
    450-463), detuning noise with a high-frequency PSD and register noise on a
    3D-capable device: pulser-core's per-trajectory samples and interaction
    matrices for a 5-atom triangular register (seed 5)."""
    from pulser_amd.pulser_adapter import problem_from_trajectory, sequence_inputs_from_pulser

    reg = Register.triangular_lattice(2, 3, spacing=6.5, prefix="q")
    seq = Sequence(reg, MockDevice)
    seq.declare_channel("ising", "rydberg_global")
    seq.add(Pulse.ConstantDetuning(BlackmanWaveform(240, 2.5), -2.0, 0.3), "ising")
    seq.add(Pulse.ConstantAmplitude(4.0, RampWaveform(160, -5.0, 6.0), 0.0), "ising")
    T = seq.get_duration()
    samples = sampler.sample(seq, extended_duration=T)
    inputs = sequence_inputs_from_pulser(samples, seq.register, seq.device)
    ext = samples.extend_duration(T + 1)
    params = dict(amp_sigma=0.08, laser_waist=20.0, temperature=30.0, trap_depth=150.0, trap_waist=1.0,
                  detuning_sigma=0.3, detuning_hf_psd=(3.0, 2.0, 1.0), detuning_hf_omegas=(10.0, 30.0, 70.0))
    nm = NoiseModel(**params)
    np.random.seed(5)
    hd = HamiltonianData(ext, seq.register, seq.device, nm, 4)
    probe = np.random.get_state()[1][:4].copy()
    qids = list(reg.qubits)
    dets, amps, inter, coords = [], [], [], []
    for traj, noisy, reps in hd.noisy_samples:
        prob = problem_from_trajectory(hd, traj, noisy, reps, 1.0)
        loc = prob["samples"]["Local"]["ground-rydberg"]
        dets.append(np.stack([loc[q]["det"] for q in range(len(qids))]))
        amps.append(np.stack([loc[q]["amp"] for q in range(len(qids))]))
        inter.append(np.asarray(prob["interaction_matrix"]))
        coords.append(np.stack([np.asarray(traj.register.qubits[q].as_array()) for q in qids]))
    print(f"waist: {len(dets)} trajectories; noise types {nm.noise_types}; amp ratio atom0 "
          f"{amps[0][0][100] / float(np.asarray(ext.samples_list[0].amp.as_array())[100]):.4f}")
    P.save_problem(os.path.join(HERE, "waist_tri6.npz"), {"inputs": inputs.to_dict()}, seed=5,
                   noise_model={k: (list(v) if isinstance(v, tuple) else v) for k, v in params.items()},
                   det=np.stack(dets), amp=np.stack(amps), interaction=np.stack(inter),
                   coords=np.stack(coords), rng_probe=probe,
                   reference_cite="pulser-core/pulser/_hamiltonian_data/hamiltonian_data.py:408-534, 758-780")


def gen_config_json():
    """``EmulationConfig.to_abstract_repr`` documents written by pulser-core
    (pulser/backend/config.py:438-447, noise_model.py:676-699, observable.py:132-139)."""
    import json
    from pulser.backend import (BitStrings, CorrelationMatrix, EmulationConfig, Energy,
                                EnergySecondMoment, EnergyVariance, Fidelity, Occupation)
    from pulser.backend.state import StateRepr

    target = StateRepr.from_state_amplitudes(eigenstates=("r", "g"), amplitudes={"rgr": 0.6, "ggg": 0.8j})
    cfgs = {
        "plain": EmulationConfig(observables=[BitStrings(num_shots=50, evaluation_times=[0.5, 1.0]),
                                              Occupation(one_state="r"), CorrelationMatrix(),
                                              Energy(tag_suffix="x"), EnergyVariance(),
                                              EnergySecondMoment(), Fidelity(target)],
                                 default_evaluation_times=[0.25, 1.0], with_modulation=True),
        "noisy": EmulationConfig(
            observables=[BitStrings()], default_evaluation_times="Full", n_trajectories=7,
            noise_model=NoiseModel(dephasing_rate=0.1, temperature=20.0, state_prep_error=0.02,
                                   p_false_pos=0.01, amp_sigma=0.05, laser_waist=150.0,
                                   eff_noise_opers=[np.array([[0, 1], [0, 0]]), np.array([[1, 0], [0, -1j]])],
                                   eff_noise_rates=[0.3, 0.1], detuning_sigma=0.2,
                                   detuning_hf_psd=(1.0, 2.0), detuning_hf_omegas=(3.0, 4.0))),
        "register": EmulationConfig(
            observables=[Occupation()], n_trajectories=3, prefer_device_noise_model=True,
            noise_model=NoiseModel(temperature=30.0, disable_doppler=True, trap_depth=150.0, trap_waist=1.0,
                                   relaxation_rate=0.2, dmm_sigma=0.1, detuning_map_spot_waist=4.0)),
    }
    from pulser.backend import Expectation
    from pulser.backend.operator import OperatorRepr

    X = {"gr": 1.0, "rg": 1.0}
    Y = {"gr": 1.0j, "rg": -1.0j}
    Z = {"rr": 1.0, "gg": -1.0}
    op = OperatorRepr.from_operator_repr(eigenstates=("r", "g"), n_qudits=3,
                                         operations=[(0.5, [(X, {0}), (Z, {1, 2})]), (2.0 - 1.0j, [(Y, {1})])])
    cfgs["operator"] = EmulationConfig(observables=[Expectation(op, evaluation_times=[1.0])])
    docs = {k: c.to_abstract_repr(skip_validation=True) for k, c in cfgs.items()}
    P.save_problem(os.path.join(HERE, "config_abstract_repr.npz"), {},
                   reference_cite="pulser-core/pulser/backend/config.py:438-470", **docs)
    for k, v in docs.items():
        print(k, json.dumps(json.loads(v))[:300])


def gen_results_json():
    """A ``pulser.backend.Results`` filled with raw values of every kind the
    default observables store, serialised by pulser-core itself
    (results.py:267-313), plus the aggregation of three such results."""
    import json
    import uuid as _uuid
    from collections import Counter
    from pulser.backend.observable import AggregationMethod
    from pulser.backend.results import Results

    def make(shift):
        res = Results(atom_order=("q0", "q1", "q2"), total_duration=1000)
        entries = [
            ("bitstrings", AggregationMethod.BAG_UNION,
             lambda t: Counter({"010": 3 + shift, "111": int(10 * t) + 1})),
            ("occupation", AggregationMethod.MEAN, lambda t: [0.1 * t + shift, 0.5, 0.25 * (1 + t)]),
            ("correlation_matrix", AggregationMethod.MEAN,
             lambda t: [[t, 0.5 + shift], [0.5 + shift, 1 - t]]),
            ("energy", AggregationMethod.MEAN, lambda t: -3.25 * t + shift),
            ("expectation", AggregationMethod.MEAN, lambda t: complex(0.5 * t, 0.125 + shift)),
            ("array", AggregationMethod.MEANSTD, lambda t: np.array([t, 2 * t + shift, 1.5])),
            ("custom_skipped", AggregationMethod.SKIP, lambda t: {"note": "x", "z": complex(1, -t)}),
        ]
        for i, (tag, method, fn) in enumerate(entries):
            uid = _uuid.UUID(int=1000 + i)
            for t in (0.5, 1.0):
                res._store_raw(uuid=uid, tag=tag, time=t, value=fn(t), aggregation_method=method)
        return res

    singles = [make(s) for s in (0, 1, 2)]
    texts = [r.to_abstract_repr(skip_validation=True) for r in singles]
    agg = Results.aggregate(singles)
    agg_text = agg.to_abstract_repr(skip_validation=True)
    P.save_problem(os.path.join(HERE, "results_abstract_repr.npz"), {},
                   reference_cite="pulser-core/pulser/backend/results.py:267-488",
                   json_texts=texts, aggregated_json=agg_text)
    print("results json:", len(texts[0]), "chars; aggregated tags:", agg.get_result_tags())
    print(json.dumps(json.loads(agg_text))[:400])


if __name__ == "__main__":
    which = sys.argv[1:] or ["rydberg", "digital", "three", "cfg1", "cfg2", "cfg3", "cfg4"]
    print("pulser", pulser.__version__)
    if "rydberg" in which:
        gen_noises_rydberg()
    if "digital" in which:
        gen_noises_digital()
    if "xy" in which:
        gen_noisy_xy()
    if "all" in which:
        gen_noises_all()
    if "three" in which:
        gen_three_atom_state()
    if "cfg1" in which:
        gen_cfg1()
    if "cfg2" in which:
        gen_cfg2(12, "cfg2_chain12_anneal")
        gen_cfg2(8, "cfg2_chain8_anneal")
    if "cfg3" in which:
        gen_cfg3_small(2, 2)
        gen_cfg3_small(2, 3)
    if "cfg4" in which:
        gen_cfg4()
    if "ns14" in which:
        gen_ns_tri14()
    if "cfg3_8" in which:
        gen_cfg3_tight(2, 4)
    if "cfg3_10" in which:
        gen_cfg3_tight(2, 5)
    if "cfg3_12" in which:
        gen_cfg3_tight(2, 6)
    if "thin_cfg3_12" in which:
        thin_sketch("cfg3_tri12_dephasing.npz")
    if "cfg3_12_to1300" in which:
        gen_cfg3_tight(2, 6, t_stop=1.3)
    if "rect16" in which:
        gen_ns_rect16()
    if "spam_all" in which:
        gen_noise_spam_all()
    if "results_noisy" in which:
        gen_results_noisy()
    if "final_state_noisy" in which:
        gen_final_state_noisy()
    if "slm_effective_size" in which:
        gen_slm_effective_size()
    if "slm_masks" in which:
        gen_slm_masks()
    if "modulation" in which:
        gen_modulation()
    if "eom_limit_det" in which:
        gen_eom_limit_det()
    if "multichannel_noise" in which:
        gen_multichannel_noise()
    if "dmm" in which:
        gen_dmm()
    if "results" in which:
        gen_results_json()
    if "waist" in which:
        gen_waist()
    if "config" in which:
        gen_config_json()
