"""-m gpu: the register-resident ket kernel (k_ket) and the split-operator master equation.

Checkers: the tight oracle fixture (12-atom anneal), the other device paths (LDS-resident
k_traj, multi-launch tiled kernels, Hermitian mesolve path) and, at full size, the exact
product-state solution of non-interacting atoms.
"""
from __future__ import annotations

import numpy as np
import pytest
from scipy.linalg import expm

from helpers import SPLIT_BUDGET, blockade_radius, load_fixture, with_anneal_samples
from pulser_amd import problem as P

pytestmark = pytest.mark.gpu

AMP_TOL = 1e-7  # amplitudes / rho entries vs the tight oracle


def _engine(probs, mode):
    from pulser_amd.engine import Engine

    return Engine.from_problems(probs, mode=mode)


def real_local_problem(n, seed=0, duration=61, collapse_ops=None, spacing=7.0):
    """Per-atom REAL drives (phase 0) and detunings: Local addressing without a phase."""
    rng = np.random.default_rng(seed)
    t = np.arange(duration) / 1000.0
    coords = P.register_coords(P.square_rect(1, n), spacing) + rng.normal(0, 0.3, (n, 2))
    z = np.zeros(duration)
    prob = P.make_ising_problem(coords, {"amp": z, "det": z, "phase": z})
    loc = {}
    for q in range(n):
        a, b, c = rng.uniform(2, 12, 3)
        loc[q] = {"amp": a * (1 + 0.5 * np.sin(2 * np.pi * (q + 1) * t / t[-1])),
                  "det": b * np.cos(3 * t + q) - c, "phase": z}
    prob["samples"] = {"Global": {}, "Local": {"ground-rydberg": loc}}
    prob["collapse_ops"] = list(collapse_ops or [])
    return prob


def tri_problem(rows, cols, collapse_ops=None):
    coords = P.register_coords(P.triangular_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=collapse_ops)


def test_ket_kernel_cfg2_chain12_against_tight_oracle_and_persistent_kernel():
    prob, extra = load_fixture("cfg2_chain12_anneal.npz")
    prob = with_anneal_samples(prob)
    times = np.asarray(extra["eval_times"])
    ref = np.asarray(extra["oracle_states_tight"])
    outs = {}
    for force in (True, False):
        with _engine([prob], "sesolve") as eng:
            eng.set_path(False, force_ket=force)
            st = eng.new_state()
            snaps = eng.solve(st, times).cpu().numpy()[:, 0]
            outs[force] = snaps
            if force:
                assert eng.stats()["n_launches"] == 1  # the whole sequence is one launch
    for k in range(1, len(times)):
        assert np.max(np.abs(outs[True][k - 1] - ref[k])) < AMP_TOL, k
    assert np.max(np.abs(outs[True] - outs[False])) < 5e-8  # two steppers, each ~1e-8 from the truth


@pytest.mark.parametrize("n", [10, 13, 14])
def test_ket_kernel_per_atom_real_drives_against_tiled_kernels(n):
    probs = [real_local_problem(n, seed=s) for s in range(2)]
    times = np.array([0.0, 0.017, 0.06])
    outs = {}
    for force in (True, False):
        with _engine(probs, "sesolve") as eng:
            eng.set_path(not force, force_ket=force, no_ket=not force)
            outs[force] = eng.solve(eng.new_state(), times).cpu().numpy()
    # two different steppers (in-place scheme at 2e-11, Taylor at 1e-10 per exponential, 120 of them)
    assert np.max(np.abs(outs[True] - outs[False])) < 2e-8
    assert abs(np.linalg.norm(outs[True][-1, 1]) - 1.0) < 1e-9


def test_ket_kernel_14_atom_triangular_anneal_against_multi_launch():
    prob = tri_problem(2, 7)
    outs = {}
    for no_ket in (False, True):
        with _engine([prob] * 8, "sesolve") as eng:  # from 8 sequences on the ket kernel is the default
            eng.set_path(False, no_ket=no_ket, no_split14=True)  # (k_ket itself: not the split-operator kernel)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.25)  # first part of the rise: Omega and |delta| both large
            outs[no_ket] = st.cpu().numpy()[0]
            if not no_ket:
                assert eng.stats()["n_launches"] == 1
    assert np.max(np.abs(outs[False] - outs[True])) < 2e-8


@pytest.mark.parametrize("n", [10, 12])
def test_split_operator_rows_against_hermitian_path(n):
    """rho(t) by Strang blocks U (D rho) U^dagger on the ket kernel against the multi-launch
    Lindbladian (k_apply14 row pass + symmetrisation), interacting atoms, per-atom drives."""
    ops = [(np.sqrt(2 * 0.05), "sigma_rr")]
    prob = real_local_problem(n, seed=3, duration=41, collapse_ops=ops)
    times = np.array([0.0, 0.013, 0.04])
    outs = {}
    for rows in (True, False):
        with _engine([prob], "mesolve") as eng:
            eng.set_path(False, force_ket=rows, no_ket=not rows, force_tile14=not rows)
            outs[rows] = eng.solve(eng.new_state(), times).cpu().numpy()[:, 0]
    rho = outs[True][-1]
    assert np.max(np.abs(outs[True] - outs[False])) < 2e-8
    assert abs(np.trace(rho).real - 1.0) < 5e-9
    assert np.max(np.abs(rho - rho.conj().T)) < 5e-12  # Hermitian up to rounding (not enforced)


def _single_atom_lindblad(omega, delta, gamma, t):
    H = np.array([[-delta, omega / 2], [omega / 2, 0.0]], dtype=complex)
    C = np.sqrt(2 * gamma) * np.diag([1.0, 0.0]).astype(complex)
    I2 = np.eye(2)
    L = (-1j * (np.kron(H, I2) - np.kron(I2, H.T)) + np.kron(C, C.conj())
         - 0.5 * np.kron(C.conj().T @ C, I2) - 0.5 * np.kron(I2, (C.conj().T @ C).T))
    return (expm(L * t) @ np.array([0, 0, 0, 1.0], dtype=complex)).reshape(2, 2)  # from |g><g|


@pytest.mark.parametrize("n, gamma", [(10, 0.5), (14, 0.05)])
def test_split_operator_rows_product_state_full_size(n, gamma):
    """Non-interacting atoms under a constant drive: rho(t) is the product of single-atom
    Lindblad solutions - an exact reference at any size (14 atoms: rho = 4.29 GB)."""
    import torch

    coords = P.register_coords(P.square_rect(1, n), 60.0)  # U ~ 1e-4 rad/us: negligible
    T = 12
    samples = {"amp": np.full(T + 1, 6.0), "det": np.full(T + 1, -2.0), "phase": np.zeros(T + 1)}
    prob = P.make_ising_problem(coords, samples, collapse_ops=[(np.sqrt(2 * gamma), "sigma_rr")])
    t_end = 0.008
    with _engine([prob], "mesolve") as eng:
        eng.set_path(False, force_ket=True)  # 10 atoms: below the default size of the row path
        st = eng.new_state()
        eng.evolve(st, 0.0, t_end)
        # 8 steps = 2 blocks of 2 + 2 steps (k_ket rows; 14 atoms at the slow rate: ONE block of 4 + 4 on k_split_reg);
        # per half-block one conjugation = 2 row passes + 1 transposition
        assert eng.stats()["n_launches"] == (1 if n == 14 else 2) * 2 * 3
        r1 = _single_atom_lindblad(6.0, -2.0, gamma, t_end)
        D = 1 << n
        rng = np.random.default_rng(1)
        pairs = [(0, 0), (D - 1, D - 1), (1, 2), (D - 1, 0), ((1 << (n - 1)) + 3, 5)]
        pairs += [tuple(int(v) for v in rng.integers(0, D, 2)) for _ in range(40)]
        got = np.array([st[0, a, b].item() for a, b in pairs])
        ref = np.array([np.prod([r1[(a >> (n - 1 - k)) & 1, (b >> (n - 1 - k)) & 1] for k in range(n)])
                        for a, b in pairs])
        diag = torch.diagonal(st[0]).real
        assert abs(float(diag.sum().item()) - 1.0) < 1e-11
        assert np.max(np.abs(got - ref)) < 2e-9  # residual interaction 1e-4 rad/us x 8 ns
        a, b = pairs[-1]
        assert abs(st[0, a, b].item() - np.conj(st[0, b, a].item())) < 1e-15
        occ = eng.occupations(st).cpu().numpy()[0]
        assert np.allclose(occ[:n], r1[0, 0].real, atol=1e-9)


# ---------------------------------------------------------------------------- Krylov (cfg5)
def rect_problem(rows, cols):
    coords = P.register_coords(P.square_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples())


def test_krylov_exponential_cfg2_chain12_against_tight_oracle():
    """BASELINE configs[4] names a Krylov-subspace sesolve: the Lanczos exponential on the tiled
    generator kernels (method='krylov') against the tight oracle of the 12-atom anneal."""
    prob, extra = load_fixture("cfg2_chain12_anneal.npz")
    prob = with_anneal_samples(prob)
    times = np.asarray(extra["eval_times"])
    ref = np.asarray(extra["oracle_states_tight"])
    with _engine([prob], "sesolve") as eng:
        snaps = eng.solve(eng.new_state(), times, method="krylov").cpu().numpy()[:, 0]
        s = eng.stats()
    for k in range(1, len(times)):
        assert np.max(np.abs(snaps[k - 1] - ref[k])) < AMP_TOL, k
    assert s["n_launches"] > s["n_applications"]  # the multi-launch path ran (dots / updates per vector)


@pytest.mark.parametrize("n, rows, cols, t1", [(16, 4, 4, 0.06), (20, 4, 5, 0.03)])
def test_krylov_against_taylor_cfg5_sizes(n, rows, cols, t1):
    """cfg5 (20-atom 4x5 register; 16 atoms as the smaller sibling): Lanczos vs the CF4 + Taylor
    stepper on the same slice of the anneal, both well inside the 1e-7 bar of each other."""
    prob = rect_problem(rows, cols)
    outs = {}
    for method in ("taylor", "krylov"):
        with _engine([prob], "sesolve") as eng:
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.02)  # leave the trivial all-ground corner first (same stepper)
            eng.evolve(st, 0.02, 0.02 + t1, method=method)
            outs[method] = st.cpu().numpy()[0]
    assert np.max(np.abs(outs["taylor"] - outs["krylov"])) < 1e-8
    assert abs(np.linalg.norm(outs["krylov"]) - 1.0) < 1e-9


def test_krylov_fused_iteration_batch_and_invariant_start():
    """The two-launch Lanczos iteration (round 6: inner products in k_apply's epilogue, k_kry_update_fused with its provisional
    scale) on a BATCH of three different 13-atom sequences (per-entry accumulators, scales and tridiagonal matrices), from
    t = 0 - where the anneal's drive is zero, the ket an eigenvector of the diagonal generator and the first Krylov vector an
    exact breakdown (u = 0 up to rounding: the stored vector is normalised noise with a true beta of ~1e-16) - against
    CF4 + Taylor; and on per-atom complex drives that start at full amplitude."""
    from helpers import chain_problem, local_problem

    base = chain_problem(13)
    g = base["samples"]["Global"]["ground-rydberg"]
    probs = []
    for f in (1.0, 0.8, 0.6):
        q = dict(base)
        q["samples"] = {"Global": {"ground-rydberg": {"amp": g["amp"] * f, "det": g["det"] * (2.0 - f), "phase": g["phase"]}}, "Local": {}}
        probs.append(q)
    for batch, t1 in ((probs, 0.15), ([local_problem(13, seed=s0, duration=101) for s0 in (3, 4)], 0.1)):
        outs = {}
        for method in ("taylor", "krylov"):
            with _engine(batch, "sesolve") as eng:
                st = eng.new_state()
                eng.evolve(st, 0.0, t1, method=method, tol=1e-12)
                outs[method] = st.cpu().numpy()
                stats = eng.stats()
        assert np.max(np.abs(outs["taylor"] - outs["krylov"])) < 1e-8
        assert np.max(np.abs(np.linalg.norm(outs["krylov"], axis=1) - 1.0)) < 1e-9
        # the last solve was the Lanczos one: apply + update per basis vector (+ a handful per exponential), not five launches each
        assert stats["n_launches"] < 3.5 * stats["n_applications"], stats


# ------------------------------------------------------------- cfg3 through the drop-in API
def test_cfg3_noise_model_end_to_end_14_atoms_keeps_density_matrices_on_the_device():
    """BASELINE configs[2] through ``QutipEmulator(...).run().sample_final_state()``
    (simulation.py:800-883, qutip_result.py:101-158, simresults.py:522-568): dephasing + SPAM
    NoiseModel on 14 atoms -> the reference picks mesolve; rho (4.29 GB) never reaches the host,
    the sampling weights are its device-reduced diagonal.  Non-interacting atoms, so the exact
    diagonal is a product of single-atom Lindblad solutions and the seeded Counter can be replayed
    with the oracle's sampling chain."""
    from oracle import sampling as osamp
    from pulser_amd import NoiseModel, QutipEmulator
    from pulser_amd.hamiltonian_data import single_global_channel
    from pulser_amd.results import DeviceState

    n, T, gamma = 14, 12, 0.05
    coords = P.register_coords(P.square_rect(1, n), 60.0)
    smp = {"amp": np.full(T, 6.0), "det": np.full(T, -2.0), "phase": np.zeros(T)}
    inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
    nm = NoiseModel(dephasing_rate=gamma, p_false_pos=0.01, p_false_neg=0.05)
    np.random.seed(11)
    sim = QutipEmulator(inputs, noise_model=nm, evaluation_times="Minimal")
    with pytest.warns(DeprecationWarning):
        res = sim.run()
    final = res.states[-1]
    assert isinstance(final, DeviceState) and final.shape == (1 << n, 1 << n)
    assert isinstance(res.states[0], DeviceState) and abs(res.states[0].tr() - 1.0) < 1e-14
    assert abs(final.tr() - 1.0) < 1e-10
    rng_state = np.random.get_state()
    counts = res.sample_final_state(N_samples=2000)
    # replay: exact product diagonal -> weights -> multinomial -> measurement flips (oracle chain).
    # The extra trailing sample (simulation.py:173) makes the spline ramp the drive down over the
    # last interval, so the single-atom reference integrates the same not-a-knot spline.
    from oracle import qutip_path as qp

    one = P.make_ising_problem(np.zeros((1, 2)), {k: np.append(v, 0.0 if k != "phase" else v[-1])
                                                    for k, v in smp.items()},
                               collapse_ops=[(np.sqrt(2 * gamma), "sigma_rr")])
    r1 = qp.mesolve(qp.build_hamiltonian(one), qp.all_ground_state(1, one["eigenbasis"]),
                    np.array([0.0, T * 1e-3]), **qp.TIGHT)[-1]
    idx = np.arange(1 << n)
    diag = np.ones(1 << n)
    for k in range(n):
        bit = (idx >> (n - 1 - k)) & 1
        diag *= np.where(bit == 0, r1[0, 0].real, r1[1, 1].real)
    assert np.max(np.abs(final.diag().real - diag)) < 5e-9
    np.random.set_state(rng_state)
    w = osamp.weights(np.diag(diag) if False else diag.astype(complex) ** 0.5, n, ["r", "g"], "ground-rydberg")
    expected = osamp.spam_flips(osamp.get_samples(w, 2000, n), 0.01, 0.05)
    assert counts == expected
    # "Full" would store 13 density matrices of 4.29 GB: still fits; 3101 of them must be refused
    with pytest.raises(MemoryError, match="evaluation_times='Minimal'"):
        sim._check_snapshot_budget(3101, 16 * 4**n)


# ------------------------------------------- double-flip dissipators on the split-operator path
from helpers import DEPOL_PAULIS  # noqa: E402

DBL_CASES = {
    "relaxation": ([(np.sqrt(0.04), "sigma_gr")], None),
    "relaxation+dephasing": ([(np.sqrt(2 * 0.05), "sigma_rr"), (np.sqrt(0.03), "sigma_gr")], None),
    "depolarizing": ([(np.sqrt(0.06 / 4), "x"), (np.sqrt(0.06 / 4), "y"), (np.sqrt(0.06 / 4), "z")], DEPOL_PAULIS),
}


@pytest.mark.parametrize("case", list(DBL_CASES))
@pytest.mark.parametrize("n", [10, 12])
def test_split_operator_rows_with_double_flip_dissipators(case, n):
    """Relaxation / depolarizing (C rho C^+ terms) on the row path: the dissipator factor is the
    product over atoms of exp(tau S_local) (k_local_exp pair passes) between the unitary row passes;
    against the multi-launch Lindbladian (three pair passes per application)."""
    ops, paulis = DBL_CASES[case]
    prob = real_local_problem(n, seed=5, duration=41, collapse_ops=ops)
    prob["depolarizing_pauli_2ds"] = dict(paulis or {})
    times = np.array([0.0, 0.011, 0.04])
    outs = {}
    for rows in (True, False):
        with _engine([prob], "mesolve") as eng:
            eng.set_path(False, no_ket=not rows)
            outs[rows] = eng.solve(eng.new_state(), times).cpu().numpy()[:, 0]
            if rows:
                s = eng.stats()
                assert s["n_launches"] < 12 * s["n_steps"], s  # not one launch set per application
    rho = outs[True][-1]
    assert np.max(np.abs(outs[True] - outs[False])) < 5e-8
    assert abs(np.trace(rho).real - 1.0) < 5e-9
    assert np.max(np.abs(rho - rho.conj().T)) < 5e-12


def test_split_operator_rows_relaxation_product_state_12_atoms():
    """Non-interacting atoms with relaxation + dephasing: exact product of single-atom solutions."""
    import torch
    from scipy.linalg import expm

    n, gam_d, gam_r = 12, 0.05, 0.2
    coords = P.register_coords(P.square_rect(1, n), 60.0)
    T = 12
    samples = {"amp": np.full(T + 1, 6.0), "det": np.full(T + 1, -2.0), "phase": np.zeros(T + 1)}
    prob = P.make_ising_problem(coords, samples, collapse_ops=[(np.sqrt(2 * gam_d), "sigma_rr"),
                                                               (np.sqrt(gam_r), "sigma_gr")])
    t_end = 0.008
    H = np.array([[2.0, 3.0], [3.0, 0.0]], dtype=complex)
    cs = [np.sqrt(2 * gam_d) * np.diag([1.0, 0.0]).astype(complex),
          np.sqrt(gam_r) * np.array([[0, 0], [1.0, 0]], dtype=complex)]  # sigma_gr = |g><r|
    I2 = np.eye(2)
    L = -1j * (np.kron(H, I2) - np.kron(I2, H.T))
    for C in cs:
        L = L + np.kron(C, C.conj()) - 0.5 * np.kron(C.conj().T @ C, I2) - 0.5 * np.kron(I2, (C.conj().T @ C).T)
    r1 = (expm(L * t_end) @ np.array([0, 0, 0, 1.0], dtype=complex)).reshape(2, 2)
    with _engine([prob], "mesolve") as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, t_end)
        D = 1 << n
        rng = np.random.default_rng(2)
        pairs = [(0, 0), (D - 1, D - 1), (1, 2), (D - 1, 0)] + [tuple(int(v) for v in rng.integers(0, D, 2)) for _ in range(40)]
        got = np.array([st[0, a, b].item() for a, b in pairs])
        ref = np.array([np.prod([r1[(a >> (n - 1 - k)) & 1, (b >> (n - 1 - k)) & 1] for k in range(n)])
                        for a, b in pairs])
        assert abs(float(torch.diagonal(st[0]).real.sum().item()) - 1.0) < 1e-11
        assert np.max(np.abs(got - ref)) < 5e-8


# ---------------------------------------------------------------------------------------------------------
# complex drives on the register-resident kernel (KET_GAUGE): c_k(t) = 0.5 Omega e^{-i phi}
# (hamiltonian.py:349-351) is gauged away inside the kernel, the state rotated back at snapshots / the end
# ---------------------------------------------------------------------------------------------------------
def _with_phases(prob, phase_fn):
    prob = dict(prob)
    loc = {q: dict(v) for q, v in prob["samples"]["Local"]["ground-rydberg"].items()}
    T = len(next(iter(loc.values()))["amp"])
    t = np.arange(T) / 1000.0
    for q in loc:
        loc[q]["phase"] = phase_fn(q, t)
    prob["samples"] = {"Global": {}, "Local": {"ground-rydberg": loc}}
    return prob


PHASES = {
    "time-dependent": lambda q, t: 0.3 * q + 0.8 * np.sin(5 * t),          # helpers.local_problem's phases
    "constant-per-atom": lambda q, t: 0.7 * q - 1.0 + 0 * t,
    "chirp": lambda q, t: 40.0 * t + 300.0 * t * t * (1 + q % 3),          # theta' up to ~76 rad/us
    "jump": lambda q, t: np.where(t < 0.03, 0.2 * q, 0.2 * q + 1.1),       # phase jump at constant amplitude
}


@pytest.mark.parametrize("kind", sorted(PHASES))
@pytest.mark.parametrize("n", [10, 13, 14])
def test_complex_drives_run_on_the_ket_kernel_in_one_launch(n, kind):
    probs = [_with_phases(real_local_problem(n, seed=s), PHASES[kind]) for s in range(2)]
    times = np.array([0.0, 0.017, 0.041, 0.06])
    outs = {}
    for force in (True, False):
        with _engine(probs, "sesolve") as eng:
            eng.set_path(not force, force_ket=force, no_ket=not force)
            outs[force] = eng.solve(eng.new_state(), times, tol=0.0 if force else 1e-12).cpu().numpy()
            if force:
                assert eng.stats()["n_launches"] == 1, "complex drives must stay on k_ket"
    assert np.max(np.abs(outs[True] - outs[False])) < 2e-8, np.max(np.abs(outs[True] - outs[False]), axis=(1, 2))
    assert abs(np.linalg.norm(outs[True][-1, 1]) - 1.0) < 1e-9


def test_complex_global_drive_starting_from_zero_amplitude_against_the_oracle():
    """A global pulse with a phase (the commonest complex drive): amplitude ramps from 0, so the gauge direction at
    t = 0 comes from the derivative; checked against the tight oracle integrated here."""
    from oracle import qutip_path as qp

    n = 10
    prob = tri_problem(2, 5)
    g = dict(prob["samples"]["Global"]["ground-rydberg"])
    T = 301
    g = {k: np.asarray(v)[:T].copy() for k, v in g.items()}
    g["amp"][-1] = 0.0
    g["phase"] = np.full(T, 0.9)
    prob = dict(prob, duration=T, samples={"Global": {"ground-rydberg": g}, "Local": {}})
    times = np.array([0.0, 0.12, 0.3])
    opts = dict(qp.default_options([np.stack([g["amp"], g["det"]])], T - 1))
    opts.update(qp.TIGHT)
    ref = qp.sesolve(qp.build_hamiltonian(prob), qp.all_ground_state(n, prob["eigenbasis"]), times, **opts)
    with _engine([prob], "sesolve") as eng:
        eng.set_path(False, force_ket=True)
        snaps = eng.solve(eng.new_state(), times).cpu().numpy()[:, 0]
        assert eng.stats()["n_launches"] == 1
    for k in (1, 2):
        assert np.max(np.abs(snaps[k - 1] - ref[k])) < AMP_TOL, k


@pytest.mark.parametrize("dphi", [0.0, 1.0, np.pi / 2])
def test_pulse_delay_phase_shifted_pulse_against_the_oracle(dphi):
    """Ramsey / echo shape (round-3 ADVICE, high): a pulse, 60 ns at zero amplitude, a second pulse whose phase is shifted
    by dphi.  While the amplitude is zero the gauged kernel holds the drive's old direction and theta' is not sampled, so
    following the direction through theta' alone would lose the step (0.50 error in the amplitudes at dphi = pi / 2 in a
    NumPy replica of the kernel logic).  The host now refuses the gauge when a series re-emerges from a sub-threshold
    stretch with another direction (compute_bounds: the complex-coefficient kernels take over); dphi = 0 stays gauged.
    10 atoms, tight oracle."""
    from oracle import qutip_path as qp

    n, T = 10, 311
    prob = tri_problem(2, 5)
    t = np.arange(T)
    amp = np.zeros(T)
    for a, b in ((0, 125), (185, 310)):
        amp[a:b + 1] = 9.0 * np.sin(np.pi * (t[a:b + 1] - a) / (b - a)) ** 2
    g = {"amp": amp, "det": np.full(T, -3.0), "phase": np.where(t < 150, 0.4, 0.4 + dphi)}
    prob = dict(prob, duration=T, samples={"Global": {"ground-rydberg": g}, "Local": {}})
    times = np.array([0.0, 0.125, 0.16, 0.31])
    opts = dict(qp.default_options([np.stack([g["amp"], g["det"]])], T - 1))
    opts.update(qp.TIGHT)
    ref = qp.sesolve(qp.build_hamiltonian(prob), qp.all_ground_state(n, prob["eigenbasis"]), times, **opts)
    with _engine([prob], "sesolve") as eng:
        eng.set_path(False, force_ket=True)
        snaps = eng.solve(eng.new_state(), times).cpu().numpy()[:, 0]
    for k in (1, 2, 3):
        assert np.max(np.abs(snaps[k - 1] - ref[k])) < AMP_TOL, (dphi, k, np.max(np.abs(snaps[k - 1] - ref[k])))


def test_a_drive_through_zero_with_a_turning_phase_is_not_gauged():
    """theta' = Im(c' conj c) / |c|^2 is unbounded where a drive passes by zero while its phase turns - e.g. a
    phase jump of almost pi at constant amplitude: the complex spline takes c from A e^{-i phi_1} to A e^{-i phi_2}
    along a chord that passes within 0.02 A of the origin.  The host refuses the gauge there (cap 4000 rad/us)
    and the multi-launch kernels (complex coefficients) take the problem (14 atoms: k_traj stops at 13)."""
    n = 14
    prob = _with_phases(real_local_problem(n, seed=4), lambda q, tt: np.where(tt < 0.03, 0.0, 3.1))
    outs = {}
    for no_ket in (False, True):
        with _engine([prob] * 8, "sesolve") as eng:
            eng.set_path(False, no_ket=no_ket)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.06)
            assert eng.stats()["n_launches"] > 1  # not the one-launch ket kernel, with or without the switch
            outs[no_ket] = st.cpu().numpy()[0]
    assert np.array_equal(outs[False], outs[True])
    assert abs(np.linalg.norm(outs[False]) - 1.0) < 1e-9


def test_emulator_noise_trajectories_with_a_pulse_phase_run_on_the_ket_kernel():
    """End to end: a 13-atom sequence whose global pulse carries a phase (complex drive coefficients,
    hamiltonian.py:349-351), amplitude + doppler noise, 16 trajectories: `QutipEmulator` lowers the batch once and
    the whole batch advances in ONE launch of k_ket<13, KET_GAUGE>; states equal the complex-coefficient kernels."""
    from pulser_amd import NoiseModel, QutipEmulator
    from pulser_amd.hamiltonian_data import single_global_channel

    n = 13
    coords = P.register_coords(P.square_rect(1, n), 8.0)
    s = {k: np.asarray(v)[:300].copy() for k, v in P.anneal_samples().items()}
    s["phase"] = np.full(300, 0.7)
    inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
    nm = NoiseModel(temperature=50.0, amp_sigma=0.05)
    outs = {}
    for env in ("", "no_ket", "default"):
        np.random.seed(4)
        emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=16, evaluation_times="Minimal")
        hd = emu._hamiltonian_data
        tables = hd.device_tables(hd.noise_trajectories, emu._sampling_rate)
        from pulser_amd.engine import Engine

        with Engine(tables, mode="sesolve") as eng:
            # (round 4: the default path of 12 - 14 atoms is the split-operator kernel, complex drives included)
            eng.set_path(False, no_ket=env == "no_ket", no_split14=env != "default")
            st = eng.new_state()
            snaps = eng.solve(st, np.asarray(emu._eval_times_array)).cpu().numpy()
            outs[env] = (snaps, eng.stats()["n_launches"])
    assert outs[""][1] == 1 and outs["no_ket"][1] >= 1
    assert np.max(np.abs(outs[""][0] - outs["no_ket"][0])) < 2e-8
    # k_split_reg<13, 5, false, false, CPLX>; 9.8e-8 before the controller checked on the amplitude ramp (test_gpu_split.py)
    assert np.max(np.abs(outs["default"][0] - outs["no_ket"][0])) < 5e-8
    assert np.max(np.abs(outs[""][0][-1, 0] - outs[""][0][-1, 5])) > 1e-3  # the trajectories differ (noise)


def test_split_operator_rows_batch_of_two_different_12_atom_registers():
    """Two DIFFERENT density matrices in one handle (other geometry, other dephasing-free drive scale): the row kernel of
    round 4 (k_split_reg<12, 5, false, ROWS>: blockIdx.z = the matrix, persistent workgroups over its rows, per-matrix
    E0 and coefficients) against the k_ket rows of round 3."""
    ops = [(float(np.sqrt(2 * 0.2)), "sigma_rr")]
    probs = [real_local_problem(12, seed=3, duration=41, collapse_ops=ops),
             real_local_problem(12, seed=8, duration=41, collapse_ops=ops, spacing=6.5)]
    outs = {}
    for rows_ket in (False, True):
        with _engine(probs, "mesolve") as eng:
            eng.set_path(False, rows_ket=rows_ket)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.04)
            outs[rows_ket] = st.cpu().numpy()
    assert np.max(np.abs(outs[False] - outs[True])) < 2e-8
    assert np.max(np.abs(outs[False][0] - outs[False][1])) > 1e-3  # the matrices really differ
    for b in range(2):
        assert abs(np.trace(outs[False][b]).real - 1.0) < 1e-10


def test_split_operator_rows_follow_the_options_and_measure_their_error():
    """the split-operator sub-steps of the row passes (k_split_reg<.., ROWS>) used to be calibrated only.  Now
    (a) a caller who asks for a tolerance tighter than the calibration, another propagator or a fixed Taylor order gets the
    polynomial rows (k_ket, whose exponentials follow ryd_opts); (b) a generator far from the calibration point - a 5-um
    chain: nearest-neighbour interaction 350 rad/us - does too (a-priori estimate c d^4 tau^4); (c) the local error of the
    unitary sub-steps is MEASURED on the heaviest row of rho (one sub-step whole against two halves) and booked in
    ryd_stats.reserved[0]; a probe far over its allowance hands the rest of the call to the polynomial rows; (d) drives with
    abrupt edges (square / EOM pulses quench the state) keep blocks of 2 + 2 knots."""
    ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr")]
    base = P.anneal_samples()
    near = P.make_ising_problem(P.register_coords(P.triangular_rect(2, 6), blockade_radius()), base, collapse_ops=ops)
    far = P.make_ising_problem(P.register_coords(P.square_rect(1, 12), 5.0), base, collapse_ops=ops)

    def run(prob, t0, t1, path=None, **opts):
        with _engine([prob], "mesolve") as eng:
            if path:
                eng.set_path(False, **path)
            st = eng.new_state()
            eng.evolve(st, t0, t1, **opts)
            return st.cpu().numpy()[0], eng.stats()

    # the physical start of the anneal: the split-operator rows run, their measured error is booked, nothing falls back
    rho_split, s_split = run(near, 0.0, 0.04)
    rho_ket, s_ket = run(near, 0.0, 0.04, {"rows_ket": True})
    assert s_split["last_order"] in (6, 10) and s_split["reserved"][3] == 0 and 0 <= s_split["reserved"][0] <= SPLIT_BUDGET
    assert s_ket["reserved"][0] == 0.0 and np.max(np.abs(rho_split - rho_ket)) < 2e-9
    # (a) options the calibrated sub-steps cannot honour -> the k_ket rows, bit for bit
    for opts in ({"tol": 1e-12}, {"method": "taylor"}, {"taylor_order": 12}):
        rho, s = run(near, 0.0, 0.02, **opts)
        ref, _ = run(near, 0.0, 0.02, {"rows_ket": True}, **opts)
        assert np.array_equal(rho, ref), opts
    # (b) strong interactions -> the k_ket rows by the a-priori estimate
    rho, s = run(far, 0.0, 0.02)
    ref, _ = run(far, 0.0, 0.02, {"rows_ket": True})
    assert np.array_equal(rho, ref) and abs(np.trace(rho).real - 1.0) < 1e-9
    # (d) a square pulse (a quench of the all-ground matrix): blocks of 2 + 2 knots instead of the 4 + 4 calibrated along the
    # adiabatic anneal (which left 1e-8 within 20 ns of a quench) - within 2e-9 of the polynomial rows at a tight tolerance
    sq = {"amp": np.concatenate([np.full(60, base["amp"].max()), [0.0]]), "det": np.concatenate([np.full(60, -20.0), [0.0]]),
          "phase": np.zeros(61)}
    square = P.make_ising_problem(P.register_coords(P.triangular_rect(2, 6), blockade_radius()), sq, collapse_ops=ops)
    rho, s = run(square, 0.0, 0.04)
    ref, _ = run(square, 0.0, 0.04, {"rows_ket": True}, tol=1e-13, magnus_tol=1e-12)
    assert s["last_order"] in (6, 10) and s["reserved"][0] <= SPLIT_BUDGET
    assert np.max(np.abs(rho - ref)) < 2e-9, np.max(np.abs(rho - ref))
