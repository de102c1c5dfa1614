"""-m gpu: the split-operator ket path (k_split: exact diagonal phases x exact single-atom rotations,
4th-order composition, step-size control) - the default propagator of two-level kets from 14 / 15
atoms on.  Checkers: the tight oracle fixtures (8- and 12-atom anneal), the CF4 + Taylor path at
sizes the oracle cannot reach, and the step-size controller's own error estimate.
"""
from __future__ import annotations

import numpy as np
import pytest

from helpers import SPLIT_BUDGET, blockade_radius, load_fixture, with_anneal_samples
from pulser_amd import problem as P

pytestmark = pytest.mark.gpu

AMP_TOL = 1e-7  # amplitudes vs the tight oracle


def _engine(probs, mode="sesolve"):
    from pulser_amd.engine import Engine

    return Engine.from_problems(probs, mode=mode)


def rect_problem(rows, cols, scale=1.0):
    coords = P.register_coords(P.square_rect(rows, cols), blockade_radius())
    smp = P.anneal_samples()
    smp = {"amp": smp["amp"] * scale, "det": smp["det"] * (2.0 - scale), "phase": smp["phase"]}
    return P.make_ising_problem(coords, smp)


def local_problem(n, seed=0, duration=61, spacing=7.0):
    """Per-atom COMPLEX drives (time-dependent phases) and detunings."""
    rng = np.random.default_rng(seed)
    t = np.arange(duration) / 1000.0
    coords = P.register_coords(P.square_rect(1, n), spacing) + rng.normal(0, 0.3, (n, 2))
    z = np.zeros(duration)
    prob = P.make_ising_problem(coords, {"amp": z, "det": z, "phase": z})
    loc = {}
    for q in range(n):
        a, b, c = rng.uniform(2, 12, 3)
        loc[q] = {"amp": a * (1 + 0.5 * np.sin(2 * np.pi * (q + 1) * t / t[-1])),
                  "det": b * np.cos(3 * t + q) - c, "phase": 0.3 * q + 2.0 * t}
    prob["samples"] = {"Global": {}, "Local": {"ground-rydberg": loc}}
    return prob


@pytest.mark.parametrize("fixture", ["cfg2_chain8_anneal.npz", "cfg2_chain12_anneal.npz"])
@pytest.mark.parametrize("fixed, no_loop", [(True, False), (False, False), (False, True)])
def test_split_operator_full_anneal_against_tight_oracle(fixture, fixed, no_loop):
    """Whole 3.1-us anneal, every evaluation time of the fixture: one sub-step per spline knot
    (`fixed`) and under the step-size controller - both an order of magnitude inside the bar.  12 atoms:
    one launch per closed run (the ket stays in registers) or, with `no_loop`, one launch per stage."""
    prob, extra = load_fixture(fixture)
    prob = with_anneal_samples(prob)
    times = np.asarray(extra["eval_times"])
    ref = np.asarray(extra["oracle_states_tight"])
    with _engine([prob]) as eng:
        eng.set_path(False, split_fixed=fixed, split_no_loop=no_loop)
        snaps = eng.solve(eng.new_state(), times, method="split").cpu().numpy()[:, 0]
        s = eng.stats()
    err = max(np.max(np.abs(snaps[k - 1] - ref[k])) for k in range(1, len(times)))
    assert err < AMP_TOL / 10, err
    # no generator applications: 6 stages per sub-step, one launch per stage (+ closings, checks)
    assert s["n_launches"] < 1.6 * s["n_applications"]
    if "chain12" in fixture and not no_loop:
        assert s["n_launches"] < s["n_applications"] / 20  # closed runs of up to 64 sub-steps per launch
    if not fixed:
        assert 0 < s["reserved"][0] <= SPLIT_BUDGET  # the controller's accumulated estimate met its target
        assert err < s["reserved"][0] * 3 + 1e-9  # ... and the estimate covers the real error


def test_split_controller_follows_the_tolerance():
    prob, extra = load_fixture("cfg2_chain12_anneal.npz")
    prob = with_anneal_samples(prob)
    ref = np.asarray(extra["oracle_states_tight"])[-1]
    t_end = float(np.asarray(extra["eval_times"])[-1])
    out = {}
    for tol in (1e-10, 1e-12):
        with _engine([prob]) as eng:
            st = eng.new_state()
            eng.evolve(st, 0.0, t_end, method="split", tol=tol)
            out[tol] = (np.max(np.abs(st.cpu().numpy()[0] - ref)), eng.stats())
    assert out[1e-12][0] < 5e-10 < AMP_TOL
    assert out[1e-12][1]["n_applications"] > 1.5 * out[1e-10][1]["n_applications"]
    assert out[1e-10][0] < AMP_TOL / 10


@pytest.mark.parametrize("n, seed", [(10, 0), (13, 3)])
def test_split_local_complex_drives_against_taylor(n, seed):
    """Per-atom complex drives and detunings (MODEL 0 physics); 13 atoms = two tilings."""
    prob = local_problem(n, seed)
    outs = {}
    for method in ("taylor", "split"):
        with _engine([prob]) as eng:
            st = eng.new_state()
            kw = {"tol": 1e-13} if method == "taylor" else {}
            eng.evolve(st, 0.0, 0.06, method=method, **kw)
            outs[method] = st.cpu().numpy()[0]
    assert np.max(np.abs(outs["taylor"] - outs["split"])) < 5e-9
    assert abs(np.linalg.norm(outs["split"]) - 1.0) < 1e-12


@pytest.mark.parametrize("n, rows, cols, ns", [(15, 3, 5, 40), (16, 4, 4, 40), (20, 4, 5, 30), (22, 2, 11, 6)])
def test_split_is_the_default_from_15_atoms_and_matches_taylor(n, rows, cols, ns):
    """cfg5 sizes (20 atoms = BASELINE configs[4]; 22 atoms = 2^13 tiles, one pass per stage):
    the default path is the split-operator one and agrees with CF4 + Taylor far inside the bar."""
    prob = rect_problem(rows, cols)
    outs = {}
    for method in ("taylor", "auto"):
        with _engine([prob]) as eng:
            st = eng.new_state()
            t0 = 1.0
            eng.evolve(st, 0.0, t0)  # a non-trivial start (the same path for both)
            eng.reset_stats()
            eng.evolve(st, t0, t0 + ns * 1e-3, method=method)
            outs[method] = st.cpu().numpy()[0]
            s = eng.stats()
            if method == "auto":
                assert s["n_applications"] >= 10 * ns // 8 and s["reserved"][0] > 0  # stages; error estimate kept
                assert s["passes"] == 1  # 21 - 22 atoms: 2^13 tiles keep one pass per stage
    assert np.max(np.abs(outs["taylor"] - outs["auto"])) < 2e-9
    assert abs(np.linalg.norm(outs["auto"]) - 1.0) < 1e-9


def test_split_batch_of_different_sequences_and_snapshots():
    """Three different 16-atom sequences in one handle, snapshots at uneven times (two inside a knot
    interval), against the Taylor path."""
    probs = [rect_problem(4, 4, s) for s in (1.0, 0.93, 1.05)]
    times = [1.0, 1.0105, 1.0107, 1.03]
    outs = {}
    for method in ("taylor", "split"):
        with _engine(probs) as eng:
            st = eng.new_state()
            eng.evolve(st, 0.0, 1.0, method="taylor")
            outs[method] = eng.solve(st, times, method=method).cpu().numpy()
    assert outs["split"].shape == (3, 3, 1 << 16)
    assert np.max(np.abs(outs["taylor"] - outs["split"])) < 2e-9
    assert np.max(np.abs(outs["split"][-1, 0] - outs["split"][-1, 1])) > 1e-3  # sequences really differ


def test_split_controller_state_survives_between_calls():
    """A front end that advances from evaluation time to evaluation time (one ryd_evolve per time) must
    not pay a controller check per call, and must end where a single call ends."""
    prob = rect_problem(3, 5)
    with _engine([prob]) as eng:
        a = eng.new_state()
        eng.evolve(a, 0.0, 1.0, method="taylor")
        b = a.clone()
        eng.reset_stats()
        eng.evolve(a, 1.0, 1.1)
        one = eng.stats()["n_launches"]
        eng.reset_stats()
        for k in range(100):
            eng.evolve(b, 1.0 + k * 1e-3, 1.0 + (k + 1) * 1e-3)
        many = eng.stats()["n_launches"]
    # (one call cuts its linear stretches into sub-steps of the working length, a hundred one-knot calls cannot: two different
    # discretisations inside the same error budget - round 6 measured 1.1e-9, before the sub-step groups 3e-10)
    assert np.max(np.abs(a.cpu().numpy() - b.cpu().numpy())) < 1e-8
    # per call: 6 stages + 1 closing pass (+ the coefficient kernel is not counted); a check costs ~20 more
    # (one-knot calls leave nothing to merge: ryd_solve picks the 6-stage scheme for them)
    assert many < 100 * 7 + 6 * 25 and one < many


def test_split_14_atoms_single_sequence_is_default_and_matches_ket_kernel():
    """One 14-atom sequence (fewer than the 8 the register-resident k_ket wants): split-operator
    passes by default, against k_ket (forced) on the north-star register."""
    coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
    prob = P.make_ising_problem(coords, P.anneal_samples())
    outs = {}
    for force in (False, True):
        with _engine([prob]) as eng:
            eng.set_path(False, force_ket=force)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.7)
            outs[force] = st.cpu().numpy()[0]
            s = eng.stats()
            assert (s["n_launches"] == 1) == force
    assert np.max(np.abs(outs[False] - outs[True])) < 2e-8


def test_split_large_ket_against_the_product_state_solution():
    """25 atoms (0.5 GiB ket, three tilings: 64-bit index arithmetic, strided tiles) on the default
    path: with the atoms far apart every amplitude is a product of single-atom amplitudes."""
    n, T = 25, 8
    coords = P.register_coords(P.square_rect(1, n), 40.0)
    samples = {"amp": np.full(T + 1, 6.0), "det": np.full(T + 1, -2.0), "phase": np.zeros(T + 1)}
    with _engine([P.make_ising_problem(coords, samples)]) as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.004)
        s = eng.stats()
        assert s["passes"] == 2 and s["n_applications"] >= 24
        h1 = np.array([[2.0, 3.0], [3.0, 0.0]])  # (r, g): -delta n_r + (Omega / 2) sigma_x
        w, v = np.linalg.eigh(h1)
        a1 = (v @ np.diag(np.exp(-1j * w * 0.004)) @ v.conj().T) @ np.array([0.0, 1.0])
        idx = [0, 1, (1 << n) - 1, (1 << (n - 1)) + 5, 0x155AAAA]
        got = st[0, idx].cpu().numpy()
        ref = np.array([np.prod([a1[(i >> (n - 1 - k)) & 1] for k in range(n)]) for i in idx])
        assert np.max(np.abs(got - ref)) < 1e-10
        import torch
        assert abs(float(torch.linalg.vector_norm(st).item()) - 1.0) < 1e-12


def test_split_controller_shrinks_the_step_for_strong_interactions():
    """Atoms at 4.5 um (nearest-neighbour interaction ~650 rad/us, 50x the drive): one sub-step per knot
    would be far too long; the controller finds that out from its own measurements (incl. restores of the
    checkpoint) and still ends on the Taylor solution."""
    n, T = 10, 201
    t = np.arange(T) / 1000.0
    coords = P.register_coords(P.square_rect(2, 5), 4.5)
    smp = {"amp": 12.0 * np.sin(np.pi * t / t[-1]) ** 2, "det": -20.0 + 200.0 * t, "phase": 1.5 * t}
    prob = P.make_ising_problem(coords, smp)
    outs = {}
    for method in ("taylor", "split"):
        with _engine([prob]) as eng:
            st = eng.new_state()
            kw = {"tol": 1e-13} if method == "taylor" else {}
            eng.evolve(st, 0.0, 0.2, method=method, **kw)
            outs[method] = st.cpu().numpy()[0]
            if method == "split":
                s = eng.stats()
    assert np.max(np.abs(outs["taylor"] - outs["split"])) < 2e-8
    assert s["n_applications"] > 2 * 6 * 200  # sub-steps shorter than a knot interval
    assert s["reserved"][0] <= SPLIT_BUDGET


def test_split_noisy_trajectories_with_hf_detuning_terms_and_per_trajectory_interactions():
    """pulser-core's noise trajectories (fixture): high-frequency detuning noise = extra detuning terms per
    (trajectory, atom) on shared series, laser-waist amplitude factors, doppler shifts and register noise
    (one interaction diagonal per trajectory) - the split-operator passes against the Taylor path."""
    from pulser_amd import NoiseModel
    from pulser_amd.engine import Engine
    from pulser_amd.hamiltonian_data import HamiltonianData, SequenceInputs

    prob, extra = load_fixture("waist_tri6.npz")
    inputs = SequenceInputs.from_dict(prob["inputs"])
    kw = dict(extra["noise_model"])
    for key in ("detuning_hf_psd", "detuning_hf_omegas"):
        kw[key] = tuple(kw[key])
    np.random.seed(5)
    hd = HamiltonianData(inputs.extend_duration(inputs.max_duration + 1), NoiseModel(**kw), 4)
    tables = hd.device_tables(hd.noise_trajectories, 1.0)
    t1 = inputs.max_duration * 1e-3
    outs = {}
    for method in ("taylor", "split"):
        with Engine(tables, mode="sesolve") as eng:
            eng.set_path(True)
            st = eng.new_state()
            eng.evolve(st, 0.0, t1, method=method, **({"tol": 1e-13} if method == "taylor" else {}))
            outs[method] = st.cpu().numpy()
    assert outs["split"].shape[0] == 4
    assert np.max(np.abs(outs["taylor"] - outs["split"])) < 5e-9
    assert np.max(np.abs(outs["split"][0] - outs["split"][1])) > 1e-3  # trajectories really differ


@pytest.mark.parametrize("n, batch, seed", [(15, 2, 1), (17, 1, 2), (18, 2, 3)])
def test_split_random_local_problems_against_taylor(n, batch, seed):
    """Random geometry, per-atom complex drives and detunings, two tilings with 3 / 5 / 6 high bits."""
    probs = [local_problem(n, seed=seed + 10 * b, duration=31, spacing=6.0) for b in range(batch)]
    outs = {}
    for method in ("taylor", "auto"):
        with _engine(probs) as eng:
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.03, method=method, **({"tol": 1e-13} if method == "taylor" else {}))
            outs[method] = st.cpu().numpy()
    assert np.max(np.abs(outs["taylor"] - outs["auto"])) < 5e-9


def test_sixteen_atom_sequence_end_to_end_through_the_emulator():
    """`QutipEmulator(...).run().sample_final_state()` on a 16-atom register (65 536 amplitudes): the
    noiseless run and a 6-trajectory amplitude + doppler + SPAM run go through the split-operator passes
    (the default beyond 14 atoms); the final state equals the Taylor path's and the seeded Counter can be
    replayed from it with the reference's sampling chain (oracle)."""
    from oracle import sampling as osamp
    from pulser_amd import NoiseModel, QutipEmulator
    from pulser_amd.hamiltonian_data import single_global_channel

    n, T = 16, 300
    coords = P.register_coords(P.square_rect(4, 4), blockade_radius())
    t = np.arange(T)
    smp = {"amp": 4 * np.pi * np.sin(np.pi * t / (T - 1)) ** 2, "det": np.linspace(-12.0, 10.0, T), "phase": np.zeros(T)}
    inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
    np.random.seed(3)
    sim = QutipEmulator(inputs, evaluation_times="Minimal")
    with pytest.warns(DeprecationWarning):
        res = sim.run()
    assert sim.last_engine_stats["n_applications"] >= 6 * T and sim.last_engine_stats["reserved"][0] > 0
    psi = np.asarray(res.states[-1]).reshape(-1)
    # the same sequence on the generator kernels
    prob = sim._current_problem
    with _engine([prob]) as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, T * 1e-3, method="taylor", tol=1e-12)
        ref = st.cpu().numpy()[0]
    assert np.max(np.abs(psi - ref)) < 5e-9
    rng_state = np.random.get_state()
    counts = res.sample_final_state(N_samples=500)
    np.random.set_state(rng_state)
    w = osamp.weights(psi, n, ["r", "g"], "ground-rydberg")
    assert counts == osamp.get_samples(w, 500, n)  # the reference's sampling chain on the device's state
    assert sum(counts.values()) == 500 and all(len(k) == n for k in counts)
    # noisy run: 6 trajectories in one batch of 16-atom kets
    np.random.seed(4)
    nm = NoiseModel(temperature=50.0, amp_sigma=0.03, p_false_pos=0.01, p_false_neg=0.03, runs=6, samples_per_run=5)
    sim = QutipEmulator(inputs, noise_model=nm, evaluation_times="Minimal")
    with pytest.warns(DeprecationWarning):
        noisy = sim.run()
    assert sum(noisy[-1].bitstring_counts.values()) == 30


@pytest.mark.parametrize("n", [21, 22, 23])
def test_large_tiles_give_21_to_23_atoms_one_pass_per_stage(n):
    """2^13-amplitude tiles (k_split_s<13>; 21 - 22 atoms): two tilings instead of three, so a stage is ONE pass over
    the ket; identical amplitudes to the 2^12 tiles (the same arithmetic in another order of the tile bits), and
    the exact product-state solution of far-apart atoms.  23 atoms stay on 2^12 tiles (two passes per stage: measured
    faster than the runtime-indexed kernel a 10 + 3 bit tiling would need)."""
    T = 6
    coords = P.register_coords(P.square_rect(1, n), 40.0)
    rng = np.random.default_rng(n)
    samples = {"amp": np.full(T + 1, 6.0), "det": np.full(T + 1, -2.0), "phase": np.zeros(T + 1)}
    prob = P.make_ising_problem(coords, samples)
    idx = [0, 1, (1 << n) - 1, (1 << (n - 1)) + 5, 0x155AAA & ((1 << n) - 1)] + [int(v) for v in rng.integers(0, 1 << n, 40)]
    outs = {}
    for small in (False, True):
        with _engine([prob]) as eng:
            eng.set_path(False, split_fixed=True, split_small_tiles=small)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.003)
            s = eng.stats()
            assert s["passes"] == (2 if small or n == 23 else 1)
            outs[small] = st[0, idx].cpu().numpy()
    assert np.max(np.abs(outs[False] - outs[True])) < 1e-14
    h1 = np.array([[2.0, 3.0], [3.0, 0.0]])
    w, v = np.linalg.eigh(h1)
    a1 = (v @ np.diag(np.exp(-1j * w * 0.003)) @ v.conj().T) @ np.array([0.0, 1.0])
    ref = np.array([np.prod([a1[(i >> (n - 1 - k)) & 1] for k in range(n)]) for i in idx])
    assert np.max(np.abs(outs[False] - ref)) < 1e-10


def test_controller_checks_on_the_amplitude_ramp_of_a_short_sequence():
    """The first 300 ns of the anneal (amplitude ramp + the first knots of the sweep) on a strongly interacting 13-atom
    chain: with a check period counted in schedule steps (48 steps of up to 9 knots) the only check of this sequence was
    the one of its first multi-knot step at the foot of the ramp, whole 9-knot sub-steps ran up the ramp, and the state
    was 9.8e-8 from a tight run with an estimate of 1.7e-8 (round 4, tools/gauge_probe.py).  The period counts knot
    intervals now and a check is due when the drive bound has grown by half: error and estimate agree."""
    n = 13
    coords = P.register_coords(P.square_rect(1, n), 8.0)
    samples = {k: np.asarray(v)[:300].copy() for k, v in P.anneal_samples().items()}
    prob = P.make_ising_problem(coords, samples)
    with _engine([prob]) as eng:
        ref = eng.new_state()
        eng.evolve(ref, 0.0, 0.3, method="taylor", tol=1e-13, magnus_tol=1e-12)
        ref = ref.cpu().numpy()[0]
    with _engine([prob]) as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.3)
        s = eng.stats()
        err = np.max(np.abs(st.cpu().numpy()[0] - ref))
    assert s["reserved"][0] > 0  # the split-operator path under its controller
    assert err < 5e-8
    assert s["reserved"][0] > 0.3 * err  # the estimate covers the error (it was a sixth of it)


def _pulse_shapes():
    t = np.arange(500)
    blackman = np.blackman(500)
    return {
        # a Blackman pulse of area 3 pi under a linear detuning ramp (the smooth rise and fall of Pulser's BlackmanWaveform)
        "blackman": {"amp": blackman * (3 * np.pi / (np.sum(blackman) * 1e-3)), "det": np.linspace(-20.0, 20.0, 500)},
        # constant drive, fast detuning sweep through resonance
        "sweep": {"amp": np.full(300, 15.0), "det": np.linspace(-60.0, 60.0, 300)},
        # two pulses with a 100-ns gap (the drive falls to zero and re-emerges)
        "two_pulses": {"amp": np.concatenate([np.full(150, 12.0), np.zeros(100), np.full(150, 20.0)]),
                       "det": np.concatenate([np.full(150, -5.0), np.zeros(100), np.full(150, 8.0)])},
    }


@pytest.mark.parametrize("shape", ["blackman", "sweep", "two_pulses"])
def test_controller_estimate_covers_the_error_on_common_pulse_shapes(shape):
    """The measured step-size control on shapes other than the anneal, 13-atom chain at 7 um (U = 46 rad/us between
    neighbours): the state stays inside the bar against a tight CF4 + Taylor run and the accumulated estimate is not
    far below the true error (the defect of the step-counted check period was an estimate six times too small)."""
    smp = _pulse_shapes()[shape]
    smp = {"amp": smp["amp"], "det": smp["det"], "phase": np.zeros(len(smp["amp"]))}
    prob = P.make_ising_problem(P.register_coords(P.square_rect(1, 13), 7.0), smp)
    t_end = len(smp["amp"]) * 1e-3
    with _engine([prob]) as eng:
        ref = eng.new_state()
        eng.evolve(ref, 0.0, t_end, method="taylor", tol=1e-13, magnus_tol=1e-12)
        ref = ref.cpu().numpy()[0]
    with _engine([prob]) as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, t_end)
        s = eng.stats()
        err = np.max(np.abs(st.cpu().numpy()[0] - ref))
    print(f"{shape}: error {err:.2e}, estimate {s['reserved'][0]:.2e}, stages {s['n_applications']}, launches {s['n_launches']}")
    assert err < 5e-8, (err, s["reserved"][:4])
    if s["reserved"][0] > 0:  # the split-operator path under its controller (a schedule with nothing to merge keeps k_ket)
        assert s["reserved"][0] > 0.3 * err or err < 2e-9, (err, s["reserved"][:4])


# ---- 6th-order scheme with multi-knot sub-steps (host_split.hpp: kSplitS10) ----

def test_sixth_order_multi_knot_substeps_against_tight_oracle():
    """12-atom anneal, every evaluation time of the fixture: the default (10-stage 6th-order composition, sub-steps
    over up to 8 knot intervals where the waveforms are one polynomial) and the round-2 scheme (6 stages, 4th
    order, one knot per sub-step) both sit an order of magnitude inside the bar; the default needs fewer stages."""
    prob, extra = load_fixture("cfg2_chain12_anneal.npz")
    prob = with_anneal_samples(prob)
    times = np.asarray(extra["eval_times"])
    ref = np.asarray(extra["oracle_states_tight"])
    out = {}
    for s6 in (False, True):
        with _engine([prob]) as eng:
            eng.set_path(False, split_s6=s6, split_no_loop=True)
            snaps = eng.solve(eng.new_state(), times, method="split").cpu().numpy()[:, 0]
            out[s6] = (max(np.max(np.abs(snaps[k - 1] - ref[k])) for k in range(1, len(times))), eng.stats())
    assert out[False][0] < AMP_TOL / 10 and out[True][0] < AMP_TOL / 10, (out[False][0], out[True][0])
    assert out[False][1]["n_steps"] < 0.5 * out[True][1]["n_steps"]  # knots removed
    assert out[False][1]["n_applications"] < 0.7 * out[True][1]["n_applications"]  # stages = passes over the ket
    assert out[False][1]["reserved"][0] <= SPLIT_BUDGET  # the controller's estimate met its target with 6th-order scaling


@pytest.mark.parametrize("t0, t1, warm", [(0.45, 0.62, False), (0.5, 0.62, True), (0.5, 0.53, True), (2.55, 2.72, False),
                                           (2.6, 2.7, True), (0.0, 0.12, True)])
def test_sixth_order_scheme_across_the_kinks_of_the_anneal(t0, t1, warm):
    """14 atoms, slices that cross / start at the kinks of the anneal (0.5 and 2.6 us), where ~25 one-knot steps (spline
    ringing) are followed by multi-knot steps: the first multi-knot step of such a run is checked by the controller
    even when a periodic check is not due.  `warm`: a first call over the same start leaves the controller's state
    behind (the case that carried a one-knot sub-step's verdict into 8-knot sub-steps: 1.4e-7)."""
    coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
    prob = P.make_ising_problem(coords, P.anneal_samples())
    with _engine([prob]) as eng:
        start = eng.new_state()
        if t0 > 0:
            eng.evolve(start, 0.0, t0, method="taylor", tol=1e-12)
        ref = start.clone()
        # (the reference's Magnus budget tightened as well: the default leaves ~3e-9 on such a slice)
        eng.evolve(ref, t0, t1, method="taylor", tol=1e-13, magnus_tol=1e-12)
        st = start.clone()
        if warm:
            eng.evolve(st, t0, t0 + 0.02, method="split")
            st = start.clone()
        eng.reset_stats()
        eng.evolve(st, t0, t1, method="split")
        s = eng.stats()
    err = float(np.max(np.abs(st.cpu().numpy() - ref.cpu().numpy())))
    assert err < 2e-9, err
    assert s["n_steps"] < 0.75 * round((t1 - t0) * 1e3) or t1 - t0 < 0.05  # knots were removed


def test_scheme_follows_the_schedule_of_the_call():
    """Evaluation times at every knot leave nothing to merge: the call runs the 6-stage scheme (one-knot sub-steps are
    cheaper with it); the same engine asked for the end state alone runs the 10-stage scheme over multi-knot
    sub-steps.  Same amplitudes either way."""
    prob = rect_problem(3, 5)
    with _engine([prob]) as eng:
        start = eng.new_state()
        eng.evolve(start, 0.0, 1.0, method="taylor")
        every = 1.0 + np.arange(0, 61) * 1e-3
        eng.reset_stats()
        snaps = eng.solve(start.clone(), every, method="split")
        s_every = eng.stats()
        eng.reset_stats()
        end = eng.solve(start.clone(), every[[0, -1]], method="split")
        s_end = eng.stats()
    assert s_every["n_steps"] == 60 and s_end["n_steps"] <= 10
    assert s_every["n_applications"] % 6 == 0 and s_end["n_applications"] % 10 == 0, (s_every, s_end)
    assert s_end["n_applications"] < 0.6 * s_every["n_applications"]  # (its checks included: 190 against 372)
    assert float((snaps[-1] - end[-1]).abs().max()) < 2e-9


# ---- k_split14_loop: 14-atom batches, one workgroup per sequence, a closed run per launch ----

def _scaled_anneals(n_seq):
    coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
    base = P.anneal_samples()
    probs = []
    for b in range(n_seq):
        f = 1.0 - 0.3 * b / max(n_seq - 1, 1)
        probs.append(P.make_ising_problem(coords, {"amp": base["amp"] * f, "det": base["det"] * (2.0 - f),
                                                   "phase": base["phase"]}))
    return probs


@pytest.mark.parametrize("t0, t1", [(0.0, 0.62), (2.4, 3.1)])
def test_register_resident_14_atom_kernel_equals_the_pass_by_pass_launches(t0, t1):
    """8 different sequences across the kinks of the anneal: the same stages in the same order, so the one-launch kernel
    (tan-form rotations, E0 from its pairwise-additive pieces, a 512-entry phase table) has to agree with the tile
    passes to rounding."""
    probs = _scaled_anneals(8)
    outs, stats = {}, {}
    for no_loop in (False, True):
        with _engine(probs) as eng:
            eng.set_path(False, split_no_loop=no_loop)
            st = eng.new_state()
            if t0 > 0:
                eng.evolve(st, 0.0, t0, method="taylor")
            eng.reset_stats()
            eng.evolve(st, t0, t1, method="split")
            outs[no_loop], stats[no_loop] = st.cpu().numpy(), eng.stats()
    assert np.max(np.abs(outs[False] - outs[True])) < 1e-12
    assert stats[False]["n_applications"] == stats[True]["n_applications"]
    assert stats[False]["n_launches"] < stats[True]["n_launches"] / 15
    if t0 == 0.0:  # (a start state from the Taylor path carries that path's norm drift)
        assert np.max(np.abs(np.linalg.norm(outs[False], axis=1) - 1.0)) < 1e-11


def test_register_resident_12_atom_kernel_against_the_passes():
    """12 atoms: 16 amplitudes per lane on 256 lanes (k_split_reg<12, 4>, round 5: four waves per sequence instead of the two
    of NR = 5 - the shape of the plain 12-atom kernel at every batch size).  The same stages in the same order as the tile
    passes: agreement to rounding at every stored time, evaluation-time snapshots (stored inside the run) included."""
    coords = P.register_coords(P.square_rect(1, 12), blockade_radius())
    base = P.anneal_samples()
    probs = []
    for b in range(8):
        f = 1.0 - 0.04 * b
        probs.append(P.make_ising_problem(coords, {"amp": base["amp"] * f, "det": base["det"] * (2.0 - f), "phase": base["phase"]}))
    times = np.array([0.0, 0.1, 0.35, 0.62])
    outs, stats = {}, {}
    for name, kw in (("loop", {}), ("passes", {"split_no_loop": True})):
        with _engine(probs) as eng:
            eng.set_path(False, **kw)
            outs[name] = eng.solve(eng.new_state(), times, method="split").cpu().numpy()
            stats[name] = eng.stats()
    assert stats["loop"]["n_applications"] == stats["passes"]["n_applications"]
    assert stats["loop"]["n_launches"] < stats["passes"]["n_launches"] / 20
    assert np.max(np.abs(outs["loop"] - outs["passes"])) < 1e-12
    assert np.max(np.abs(np.linalg.norm(outs["loop"][-1], axis=1) - 1.0)) < 1e-11
    assert np.max(np.abs(outs["loop"][:, 0] - outs["loop"][:, 7])) > 1e-3  # the sequences really differ


def test_register_resident_14_atom_kernel_complex_drives_and_per_sequence_interactions():
    """Per-atom complex, time-dependent drives and a different geometry (interaction diagonal) per sequence: the
    general rotation branch and the per-sequence E0 pieces, against the tile passes and against CF4 + Taylor."""
    probs = [local_problem(14, seed=3 + 10 * b, duration=41, spacing=6.5) for b in range(8)]
    outs = {}
    for name, kw, method, opts in (("loop", {}, "split", {}), ("passes", {"split_no_loop": True}, "split", {}),
                                   ("taylor", {"no_ket": True}, "taylor", {"tol": 1e-13})):
        with _engine(probs) as eng:
            eng.set_path(False, **kw)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.04, method=method, **opts)
            outs[name] = st.cpu().numpy()
    assert np.max(np.abs(outs["loop"] - outs["passes"])) < 1e-12
    assert np.max(np.abs(outs["loop"] - outs["taylor"])) < 5e-9
    assert np.max(np.abs(outs["loop"][0] - outs["loop"][1])) > 1e-3  # the sequences really differ


def test_14_atom_batches_choose_their_kernel_from_the_schedule_of_the_call():
    """End state of the anneal: multi-knot steps -> k_split_reg (a launch per closed run, a third of k_ket's
    stages).  Evaluation times at every knot (evaluation_times="Full", the reference's default, simulation.py:137): since
    round 5 the snapshots are stored from the registers INSIDE the runs (k_split_reg<.., SNAP> + k_split_snap_close), so
    the call stays on the split kernel - 6-stage one-knot sub-steps, a launch per 64 knots - instead of k_ket; with the
    round-4 hook (`snaps_outside`) nothing merges -> k_ket, the whole schedule in one launch.  `no_split14` keeps k_ket in
    either case.  Same kets (both inside the bar by the full-size tests)."""
    probs = _scaled_anneals(8)
    res = {}
    with _engine(probs) as eng:
        for name, kw in (("auto", {}), ("k_ket", {"no_split14": True})):
            eng.set_path(False, **kw)
            st = eng.new_state()
            eng.reset_stats()
            eng.evolve(st, 0.0, 0.9)
            res[name] = (st.cpu().numpy(), eng.stats())
        every = np.arange(0, 61) * 1e-3
        snaps = {}
        for name, kw in (("inside", {}), ("outside", {"snaps_outside": True})):
            eng.set_path(False, **kw)
            eng.reset_stats()
            snaps[name] = (eng.solve(eng.new_state(), every).cpu().numpy(), eng.stats())
    assert res["k_ket"][1]["n_launches"] == 1 and snaps["outside"][1]["n_launches"] == 1
    assert res["auto"][1]["n_launches"] > 1 and res["auto"][1]["reserved"][0] > 0
    assert res["auto"][1]["n_applications"] < 0.5 * res["k_ket"][1]["n_applications"]
    assert np.max(np.abs(res["auto"][0] - res["k_ket"][0])) < 2e-8
    s_in = snaps["inside"][1]
    assert s_in["reserved"][0] > 0 and 1 < s_in["n_launches"] <= 16  # the controller booked; runs (+ its early checks), not one per evaluation time
    assert s_in["n_applications"] < 0.6 * snaps["outside"][1]["n_applications"]
    assert np.max(np.abs(snaps["inside"][0] - snaps["outside"][0])) < 2e-8  # every stored time, every sequence


@pytest.mark.parametrize("n", [12, 13, 14])
def test_snapshots_stored_inside_a_run_equal_a_closed_run_per_evaluation_time(n):
    """k_split_reg<.., SNAP> stores the OPEN state of a sub-step boundary (the last D of the sub-step is fused into the
    next stage, the cosines of the tan-form rotations ride on it) and k_split_snap_close finishes the stored kets: the same
    arithmetic as a run that closes at the evaluation time, so the two agree to rounding at EVERY stored time - dense
    times (one-knot 6-stage sub-steps), every 10th knot (6th-order multi-knot sub-steps cut by the controller) and ragged
    lists with times between knots; different sequences per batch entry; a pulse phase (the gauge of the real kernel:
    the stored frame must be the laboratory frame).  Against CF4 + Taylor at a tight tolerance too."""
    coords = P.register_coords(P.triangular_rect(2, 7) if n == 14 else P.square_rect(1, n), blockade_radius())
    base = P.anneal_samples()
    T = 901
    probs = []
    for b in range(3):
        f = 1.0 - 0.15 * b
        probs.append(P.make_ising_problem(coords, {"amp": base["amp"][:T] * f, "det": base["det"][:T] * (2.0 - f),
                                                   "phase": np.full(T, 0.4 * b)}))
    grid = np.arange(T) * 1e-3
    cases = {"dense": grid[:260], "every10": grid[::10],
             "ragged": np.unique(np.concatenate([grid[::37], grid[480:530], [0.4005, 0.51234, 0.9]]))}
    for label, times in cases.items():
        outs, stats = {}, {}
        for name, kw, opts in (("inside", {}, {}), ("outside", {"snaps_outside": True}, {"method": "split"}),
                               ("taylor", {}, {"method": "taylor", "tol": 1e-12})):
            with _engine(probs) as eng:
                eng.set_path(False, **kw)
                outs[name] = eng.solve(eng.new_state(), times, **opts).cpu().numpy()
                stats[name] = eng.stats()
        assert stats["inside"]["reserved"][0] > 0, label  # the split-operator path ran
        # runs (+ the controller's checks), not a run per evaluation time
        assert stats["inside"]["n_launches"] < stats["outside"]["n_launches"], (label, stats["inside"], stats["outside"])
        if label == "dense":
            assert stats["inside"]["n_launches"] < 0.4 * len(times), (label, stats["inside"])
        assert np.max(np.abs(outs["inside"] - outs["outside"])) < 1e-12, label
        # (the controller's budget for a whole sequence is SPLIT_BUDGET; these are 0.9 us of three differently scaled anneals)
        # the bar, and the controller's booked estimate covers the true error (round 6: the controller spends its budget on
        # the linear stretches - 5.7e-8 on the ragged list with an estimate inside the budget; until round 5 the 9-knot steps
        # left most of it unused and this read < 2e-8)
        err = float(np.max(np.abs(outs["inside"] - outs["taylor"])))
        assert err < 1e-7 and err <= max(4 * stats["inside"]["reserved"][0], 2e-9), (label, err, stats["inside"]["reserved"])
        assert stats["inside"]["reserved"][0] <= SPLIT_BUDGET, label
        assert np.max(np.abs(outs["inside"][:, 0] - outs["inside"][:, 1])) > 1e-3  # the sequences really differ


def test_modulated_local_complex_drives_take_the_split_kernel_by_default():
    """Per-atom complex drives with time-dependent phases (local addressing; nothing in the waveforms to merge): since
    round 4 the phase of a drive is carried by the D factors (SplitRun.gauge: the rotation by c = |c| e^{i theta} is
    Z R(|c|) Z^+ with a diagonal Z, exact), the real tan-form kernel runs them, and the 6-stage composition with one-knot
    sub-steps is the default at 12 - 14 atoms for calls without an evaluation time at every knot.  Against the
    polynomial kernels (complex arithmetic, k_traj / gauged k_ket) on the same batch."""
    from helpers import local_problem

    for n in (12, 13):
        probs = [local_problem(n, seed=s, duration=201) for s in range(4)]
        res = {}
        for name, kw in (("auto", {}), ("poly", {"no_split14": True})):
            with _engine(probs) as eng:
                eng.set_path(False, **kw)
                st = eng.new_state()
                eng.evolve(st, 0.0, 0.2)
                res[name] = (st.cpu().numpy(), eng.stats())
        assert res["auto"][1]["reserved"][0] > 0 and res["poly"][1]["reserved"][0] == 0  # controller booked / not
        assert res["auto"][1]["n_applications"] < 0.7 * res["poly"][1]["n_applications"]
        assert np.max(np.abs(res["auto"][0] - res["poly"][0])) < 2e-8


# ---- k_split_reg<.., CPLX>: complex drives (a pulse with a phase) on the register-resident split-operator kernel ----

@pytest.mark.parametrize("n_cols", [6, 7])
def test_anneal_with_a_pulse_phase_takes_the_register_resident_split_kernel_against_the_oracle(n_cols):
    """The anneal with a constant pulse phase (complex drive coefficients, hamiltonian.py:349-351): the waveforms are
    still one polynomial across knots, so the default path of a 12- / 14-atom register is the 6th-order split-operator
    kernel with complex rotations (k_split_reg<N, 5, false, false, true>: 4 FMAs per amplitude and bit).  First 0.62 us
    (across the kink at 0.5 us) against the tight oracle integrated here, and against the gauged polynomial kernel."""
    from oracle import qutip_path as qp

    n = 2 * n_cols
    coords = P.register_coords(P.triangular_rect(2, n_cols), blockade_radius())
    base = P.anneal_samples()
    T = 621
    g = {"amp": base["amp"][:T].copy(), "det": base["det"][:T].copy(), "phase": np.full(T, 0.9)}
    prob = P.make_ising_problem(coords, g)
    times = np.array([0.0, 0.3, 0.62])
    outs, stats = {}, {}
    for name, kw in (("default", {}), ("polynomial", {"no_split14": True})):
        with _engine([prob] * 8) as eng:
            eng.set_path(False, **kw)
            outs[name] = eng.solve(eng.new_state(), times).cpu().numpy()[:, 0]
            stats[name] = eng.stats()
    assert stats["default"]["reserved"][0] > 0 and 1 < stats["default"]["n_launches"] < stats["default"]["n_applications"] / 20
    assert stats["polynomial"]["n_launches"] == 1
    gap = float(np.max(np.abs(outs["default"] - outs["polynomial"])))
    # (a 0.62-us sequence owns the whole budget: round 6 measured 3.4e-8 / 4.1e-8 here under the largest-entry controller, estimate inside the budget)
    assert gap < 1e-7 and gap <= max(4 * stats["default"]["reserved"][0], 2e-8), (gap, stats["default"]["reserved"])
    assert stats["default"]["reserved"][0] <= SPLIT_BUDGET
    if n == 12:  # (the 14-atom oracle takes minutes: the 12-atom one pins the complex rotations)
        opts = dict(qp.default_options([np.stack([g["amp"], g["det"]])], T - 1))
        opts.update(qp.TIGHT)
        ref = qp.sesolve(qp.build_hamiltonian(prob), qp.all_ground_state(n, prob["eigenbasis"]), times, **opts)
        for k in (1, 2):
            assert np.max(np.abs(outs["default"][k - 1] - ref[k])) < 1e-7, k
