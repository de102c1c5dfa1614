"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Tolerances: amplitudes / rho entries <= 1e-7 max-abs against the
TIGHT oracle (zvode rtol 1e-13); single generator applications <= 1e-11
relative; sampling indices bit-exact (tests/test_gpu_emulator.py, tests/test_gpu_simresults.py).
"""
import numpy as np
import pytest

from helpers import (DEPOL_PAULIS, chain_problem, load_fixture, local_problem,
                     rand_state, with_anneal_samples)

pytestmark = pytest.mark.gpu

AMP_TOL = 1e-7


def _engine(problems, mode=None, **kw):
    from pulser_amd.engine import Engine

    return Engine.from_problems(problems, mode=mode, **kw)


def _to_dev(eng, host):
    import torch

    return torch.from_numpy(np.ascontiguousarray(host)).to(eng.device)


@pytest.mark.parametrize("n,tile_bits", [(3, 0), (8, 0), (8, 5), (12, 0), (14, 0), (13, 6)])
def test_generator_sesolve_global(n, tile_bits):
    from oracle import qutip_path as qp

    prob = chain_problem(n)
    ham = qp.build_hamiltonian(prob)
    eng = _engine([prob], tile_bits=tile_bits)
    x = rand_state(2**n, 1)
    for t in (0.0, 0.4321, 1.7005, 3.1):
        ref = -1j * ham.apply(t, x)
        got = eng.apply_generator(_to_dev(eng, x[None, :]), t).cpu().numpy()[0]
        assert np.max(np.abs(got - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("n,tile_bits", [(5, 0), (9, 4), (13, 0)])
def test_generator_sesolve_local_batch(n, tile_bits):
    from oracle import qutip_path as qp

    probs = [local_problem(n, seed=s) for s in range(3)]
    probs[1]["bad_atoms"][1] = True
    for k in ("amp", "det", "phase"):
        probs[1]["samples"]["Local"]["ground-rydberg"][1][k] *= 0.0
    from pulser_amd.problem import interaction_matrix, C6_LEVEL70
    probs[1]["interaction_matrix"] = interaction_matrix(probs[1]["coords"], C6_LEVEL70, probs[1]["bad_atoms"])
    eng = _engine(probs, tile_bits=tile_bits)
    xs = np.stack([rand_state(2**n, 10 + s) for s in range(3)])
    for t in (0.0, 0.1234, 0.4):
        got = eng.apply_generator(_to_dev(eng, xs), t).cpu().numpy()
        for b, p in enumerate(probs):
            ref = -1j * qp.build_hamiltonian(p).apply(t, xs[b])
            assert np.max(np.abs(got[b] - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref)))


ME_CASES = {
    "dephasing": ([(np.sqrt(0.1), "sigma_rr")], {}),
    "relaxation": ([(np.sqrt(0.07), "sigma_gr")], {}),
    "depolarizing": ([(np.sqrt(0.05 / 4), p) for p in "xyz"], DEPOL_PAULIS),
    "all": ([(np.sqrt(0.1), "sigma_rr"), (np.sqrt(0.07), "sigma_gr")] + [(np.sqrt(0.05 / 4), p) for p in "xyz"], DEPOL_PAULIS),
}


@pytest.mark.parametrize("case", list(ME_CASES))
@pytest.mark.parametrize("n,tile_bits", [(2, 0), (4, 0), (5, 6), (7, 0), (7, 7)])
def test_generator_mesolve(case, n, tile_bits):
    from oracle import qutip_path as qp

    ops, paulis = ME_CASES[case]
    prob = local_problem(n, seed=3, collapse_ops=ops, paulis=paulis)
    ham = qp.build_hamiltonian(prob)
    rhs = qp.lindblad_rhs(ham)
    eng = _engine([prob], mode="mesolve", tile_bits=tile_bits)
    D = 2**n
    rng = np.random.default_rng(5)
    x = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))  # not Hermitian on purpose
    for t in (0.0, 0.2345):
        ref = rhs(t, x.ravel()).reshape(D, D)
        got = eng.apply_generator(_to_dev(eng, x[None]), t).cpu().numpy()[0]
        assert np.max(np.abs(got - ref)) <= 1e-11 * max(1.0, np.max(np.abs(ref)))


def _evolve_check(prob, extra, mode, key_states="oracle_states_tight", psi0=None, tol=AMP_TOL):
    eng = _engine([prob], mode=mode)
    times = np.asarray(extra["eval_times"], dtype=float)
    ref = np.asarray(extra[key_states])
    state = eng.new_state(psi0)
    worst = 0.0
    for i in range(1, len(times)):
        eng.evolve(state, times[i - 1], times[i])
        got = state.cpu().numpy()[0]
        worst = max(worst, float(np.max(np.abs(got - ref[i]))))
    assert worst <= tol, worst
    return eng, state, worst


def test_evolve_cfg2_chain8():
    prob, extra = load_fixture("cfg2_chain8_anneal.npz")
    _evolve_check(with_anneal_samples(prob), extra, "sesolve")


def test_evolve_cfg2_chain12():
    prob, extra = load_fixture("cfg2_chain12_anneal.npz")
    eng, state, worst = _evolve_check(with_anneal_samples(prob), extra, "sesolve")
    # QuTiP-default tolerances are themselves ~1e-3 off at N=12 (norm drift -8e-4)
    dflt = np.asarray(extra["oracle_states_default"])[-1]
    assert np.max(np.abs(state.cpu().numpy()[0] - dflt)) < 5e-3
    occ = eng.occupations(state).cpu().numpy()[0]
    assert abs(occ[-1] - 1.0) < 1e-7  # Taylor truncation at tol 1e-10/step is not exactly unitary


@pytest.mark.parametrize("name", ["cfg3_tri4_dephasing.npz", "cfg3_tri6_dephasing.npz"])
def test_evolve_cfg3_small(name):
    prob, extra = load_fixture(name)
    eng, state, worst = _evolve_check(with_anneal_samples(prob), extra, "mesolve")
    rho = state.cpu().numpy()[0]
    assert abs(np.trace(rho) - 1.0) < 1e-9
    assert np.max(np.abs(rho - rho.conj().T)) < 1e-10


def test_evolve_cfg1_and_three_atom():
    prob, extra = load_fixture("cfg1_square4_pi.npz")
    idx = np.asarray(extra["oracle_state_indices"])
    times = np.asarray(extra["aux"]["eval_times"])[idx]
    ext = {"eval_times": times, "oracle_states_tight": extra["oracle_states_tight"]}
    _evolve_check(prob, ext, "sesolve")
    prob, extra = load_fixture("three_atom_state.npz")
    idx = np.asarray(extra["oracle_state_indices"])
    times = np.asarray(extra["aux"]["eval_times"])[idx]
    ext = {"eval_times": times, "oracle_states_tight": extra["oracle_states_tight"]}
    eng, state, _ = _evolve_check(prob, ext, "sesolve", psi0=np.asarray(extra["initial_state"]))


def test_evolve_cfg4_trajectories_batched():
    """Three noisy trajectories (doppler + amplitude + bad atoms) in one batch."""
    prob, extra = load_fixture("cfg4_chain12_noise.npz")
    from oracle import qutip_path as qp

    # only trajectory 0 is stored in full; check it alone and inside a batch of clones
    times = np.asarray(extra["eval_times"])
    ref = np.asarray(extra["oracle_states_tight"])[0]
    eng = _engine([prob, prob], mode="sesolve")
    state = eng.new_state()
    for i in range(1, len(times)):
        eng.evolve(state, times[i - 1], times[i])
        got = state.cpu().numpy()
        assert np.max(np.abs(got[0] - ref[i])) <= AMP_TOL
        assert np.array_equal(got[0], got[1])


def test_probabilities_and_occupations():
    n = 10
    prob = chain_problem(n)
    eng = _engine([prob, prob])
    xs = np.stack([rand_state(2**n, 3), rand_state(2**n, 4)])
    dev = _to_dev(eng, xs)
    w = eng.probabilities(dev, reverse=True).cpu().numpy()
    assert np.array_equal(w, (np.abs(xs) ** 2)[:, ::-1]) or np.max(np.abs(w - (np.abs(xs) ** 2)[:, ::-1])) < 1e-17
    occ = eng.occupations(dev).cpu().numpy()
    idx = np.arange(2**n)
    for b in range(2):
        p = np.abs(xs[b]) ** 2
        for k in range(n):
            nk = 1 - ((idx >> (n - 1 - k)) & 1)
            assert abs(occ[b, k] - np.sum(p * nk)) < 1e-13
        assert abs(occ[b, n] - 1.0) < 1e-13


def test_ket_to_dm_and_outer_accumulate():
    import torch

    n = 5
    prob = local_problem(n, seed=1, collapse_ops=[(0.3, "sigma_rr")])
    eng = _engine([prob, prob, prob], mode="mesolve")
    xs = np.stack([rand_state(2**n, s) for s in range(3)])
    rho = eng.new_state(xs).cpu().numpy()
    for b in range(3):
        assert np.max(np.abs(rho[b] - np.outer(xs[b], xs[b].conj()))) < 1e-16
    acc = torch.zeros((2**n, 2**n), dtype=torch.complex128, device=eng.device)
    w = np.array([1.0, 2.0, 0.5])
    eng.outer_accumulate(_to_dev(eng, xs), acc, w)
    ref = sum(w[b] * np.outer(xs[b], xs[b].conj()) for b in range(3))
    assert np.max(np.abs(acc.cpu().numpy() - ref)) < 1e-15


@pytest.mark.parametrize("n", [1, 3, 6, 8, 9, 10, 11, 12, 13])
def test_persistent_kernel_matches_generic_and_oracle(n):
    """The LDS-resident trajectory kernel (one launch) against the tiled path
    (one launch per Taylor stage) and the oracle, with per-qubit coefficients."""
    from oracle import qutip_path as qp

    probs = [local_problem(n, seed=s, duration=61) for s in range(2)]
    times = np.array([0.0, 0.0105, 0.0105, 0.03, 0.06])
    outs = {}
    for force in (False, True):
        eng = _engine(probs)
        eng.set_path(force, no_split14=True)  # the persistent kernel itself (round 4: 12 - 13 atoms take k_split_reg by default)
        st = eng.new_state()
        snaps = eng.solve(st, times).cpu().numpy()
        assert np.array_equal(snaps[-1], st.cpu().numpy())
        outs[force] = snaps
        if not force:
            assert eng.stats()["n_launches"] == 2  # split at the duplicated time
    assert np.max(np.abs(outs[False] - outs[True])) < 1e-13
    assert np.array_equal(outs[False][0], outs[False][1])
    for b, p in enumerate(probs):
        ham = qp.build_hamiltonian(p)
        ref = qp.sesolve(ham, qp.all_ground_state(n, p["eigenbasis"]), times[[0, 1, 3, 4]], max_step=1e-3, **qp.TIGHT)
        for i, j in ((0, 1), (2, 2), (3, 3)):
            assert np.max(np.abs(outs[False][i][b] - ref[j])) < AMP_TOL


def test_persistent_kernel_cfg2_chain12_snapshots():
    prob, extra = load_fixture("cfg2_chain12_anneal.npz")
    eng = _engine([with_anneal_samples(prob)])
    eng.set_path(False, no_split14=True)  # k_traj itself (round 4: 12 - 14 atoms may take k_split_reg by default)
    st = eng.new_state()
    snaps = eng.solve(st, np.asarray(extra["eval_times"])).cpu().numpy()
    ref = np.asarray(extra["oracle_states_tight"])
    assert eng.stats()["n_launches"] == 1
    for i in range(1, len(ref)):
        assert np.max(np.abs(snaps[i - 1][0] - ref[i])) < AMP_TOL


@pytest.mark.parametrize("single_pass", [False, True])
@pytest.mark.parametrize("mode,n,batch", [("sesolve", 13, 1), ("sesolve", 17, 1), ("sesolve", 13, 300),
                                          ("mesolve", 6, 1), ("mesolve", 7, 1), ("mesolve", 9, 1)])
def test_tiled_kernel_workgroup_widths_and_pass_plans(mode, n, batch, single_pass):
    """The tiled kernel in both workgroup widths (1024 threads when a launch has
    at most 512 tiles, 512 otherwise), with balanced multi-pass plans and with the
    single-launch plan of states up to 128 MiB (low bits in LDS, the partner of every
    higher bit read from the same offset of another tile), against the oracle's
    sparse matvec / Lindblad right-hand side."""
    from oracle import qutip_path as qp

    ops = [(np.sqrt(0.1), "sigma_rr")] if mode == "mesolve" else None
    prob = local_problem(n, seed=1, duration=21, collapse_ops=ops)
    eng = _engine([prob] * batch, mode=mode)
    eng.set_path(True, no_tile14=True, no_single_pass=not single_pass)
    shape = eng.state_shape
    rng = np.random.default_rng(0)
    x = rng.normal(size=shape[1:]) + 1j * rng.normal(size=shape[1:])
    dev = _to_dev(eng, np.broadcast_to(x, shape).copy())
    got = eng.apply_generator(dev, 0.0123).cpu().numpy()
    # 13-atom kets: two 2^12-tile passes, or one pass of 2^13 tiles when >= 128 tiles remain;
    # batches up to 128 MiB (these kets, 7- and 9-atom density matrices): one launch
    multi = {13: 2, 17: 2, 6: 1, 7: 2, 9: 2}[n] if batch < 128 else 1
    small = 16 * batch * (2**n if mode == "sesolve" else 4**n) <= 128 << 20
    assert eng.stats()["passes"] == (1 if single_pass and small else multi)
    ham = qp.build_hamiltonian(prob)
    if mode == "sesolve":
        ref = -1j * ham.apply(0.0123, x)
    else:
        ref = qp.lindblad_rhs(ham)(0.0123, x.ravel()).reshape(x.shape)
    scale = np.max(np.abs(ref))
    for b in (0, batch - 1):
        assert np.max(np.abs(got[b] - ref)) <= 1e-11 * scale


@pytest.mark.parametrize("n", [7, 8, 9])
def test_hermitian_mesolve_path_matches_generic_passes_and_oracle(n):
    """mesolve with a diagonal dissipator on N >= 7: the register-tile row pass
    + Hermitian symmetrisation (2 launches per application) against the generic
    3-pass tiling, and against the tight oracle at N = 7."""
    from oracle import qutip_path as qp

    ops = [(np.sqrt(0.1), "sigma_rr")]
    probs = [local_problem(n, seed=s, duration=31, collapse_ops=ops) for s in range(2)]
    times = np.array([0.0, 0.011, 0.03])
    outs = {}
    for no14 in (False, True):
        eng = _engine(probs, mode="mesolve")
        eng.set_path(False, no_tile14=no14, force_tile14=not no14)
        st = eng.new_state()
        outs[no14] = eng.solve(st, times).cpu().numpy()
        s = eng.stats()
        assert s["n_launches"] == (2 if not no14 else s["passes"]) * s["n_applications"]
    assert np.max(np.abs(outs[False] - outs[True])) < 1e-12
    rho = outs[False][-1]
    assert np.max(np.abs(rho - np.conj(np.swapaxes(rho, 1, 2)))) == 0.0  # exactly Hermitian
    if n == 7:
        for b, p in enumerate(probs):
            ham = qp.build_hamiltonian(p)
            ref = qp.mesolve(ham, qp.all_ground_state(n, p["eigenbasis"]), times[[0, 2]],
                             max_step=1e-3, **qp.TIGHT)[-1]
            assert np.max(np.abs(rho[b] - ref)) < AMP_TOL


def test_register_tile_kernel_sesolve_14_and_16_atoms():
    """k_apply14 as pass 0 (N = 14: the only pass) against the generic tiling."""
    for n in (14, 16):
        probs = [local_problem(n, seed=4, duration=21)]
        res = {}
        for no14 in (False, True):
            eng = _engine(probs)
            eng.set_path(False, no_tile14=no14, force_tile14=not no14)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.003, method="taylor")  # the generator kernels (default here: split-operator)
            res[no14] = st.cpu().numpy()
            # without the register tiles: the single-launch plan of small states
            assert eng.stats()["passes"] == ({14: 1, 16: 2}[n] if not no14 else 1)
        assert np.max(np.abs(res[False] - res[True])) < 1e-13


@pytest.mark.parametrize("n,batch", [(3, 7), (6, 5), (7, 33), (9, 130)])
def test_trajectory_averaged_density_matrix_kernels(n, batch):
    """rho += sum_t w_t |psi_t><psi_t| (density_matrix_aggregator,
    pulser_simulation/aggregators.py:20-37): the plain kernel (N < 6) and the
    fp64 matrix-core kernel (upper-triangle tiles + mirror, ragged trajectory
    counts, weights, accumulation into a non-zero matrix) against NumPy."""
    prob = local_problem(n, seed=1, duration=21)
    eng = _engine([prob] * batch, mode="sesolve")
    rng = np.random.default_rng(n)
    D = 2**n
    psi = rng.normal(size=(batch, D)) + 1j * rng.normal(size=(batch, D))
    w = rng.uniform(0.2, 2.0, batch)
    start = rng.normal(size=(D, D)) + 1j * rng.normal(size=(D, D))
    acc = _to_dev(eng, start.copy())
    dev = _to_dev(eng, psi)
    eng.outer_accumulate(dev, acc, w)
    eng.outer_accumulate(dev, acc)
    ref = start + np.einsum("t,ta,tb->ab", w + 1.0, psi, psi.conj())
    got = acc.cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref))
    pure = got - start
    assert np.max(np.abs(pure - pure.conj().T)) <= 1e-12 * np.max(np.abs(ref))


def test_large_ket_against_the_product_state_solution():
    """25 atoms (0.5 GiB ket, three tiled passes per application, 64-bit index
    arithmetic): with the atoms far apart every atom evolves on its own, so any
    amplitude is a product of single-atom amplitudes.  (tools/big_ket.py runs the
    same check at 28 and 30 atoms = 4 and 16 GiB kets.)"""
    from pulser_amd import problem as P

    n, T = 25, 8
    coords = P.register_coords(P.square_rect(1, n), 40.0)
    samples = {"amp": np.full(T + 1, 6.0), "det": np.full(T + 1, -2.0), "phase": np.zeros(T + 1)}
    eng = _engine([P.make_ising_problem(coords, samples)], mode="sesolve")
    st = eng.new_state()
    eng.evolve(st, 0.0, 0.004, tol=1e-14, method="taylor")  # exact reference: tighter than the default 1e-10
    assert eng.stats()["passes"] == 3
    h1 = np.array([[2.0, 3.0], [3.0, 0.0]])  # (r, g): -delta n_r + (Omega / 2) sigma_x
    w, v = np.linalg.eigh(h1)
    a1 = (v @ np.diag(np.exp(-1j * w * 0.004)) @ v.conj().T) @ np.array([0.0, 1.0])
    idx = [0, 1, (1 << n) - 1, (1 << (n - 1)) + 5, 0x155AAAA]
    got = st[0, idx].cpu().numpy()
    ref = np.array([np.prod([a1[(i >> (n - 1 - k)) & 1] for k in range(n)]) for i in idx])
    assert np.max(np.abs(got - ref)) < 1e-12
    import torch
    assert abs(float(torch.linalg.vector_norm(st).item()) - 1.0) < 1e-13
    eng.close()


@pytest.mark.parametrize("case", list(ME_CASES))
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
def test_persistent_density_matrix_kernel_matches_tiled_kernels_and_oracle(case, n):
    """mesolve of small registers (rho = 4 ... 4096 entries): the persistent
    one-launch kernel against the tiled kernels (one launch per Taylor stage) and,
    up to 4 atoms, the tight oracle - dephasing, relaxation and depolarizing
    (double-flip) dissipators, per-qubit drives, snapshots at evaluation times."""
    from oracle import qutip_path as qp

    ops, paulis = ME_CASES[case]
    probs = [local_problem(n, seed=s, duration=41, collapse_ops=ops, paulis=paulis) for s in range(3)]
    times = np.array([0.0, 0.0105, 0.0105, 0.04])
    outs = {}
    for force in (False, True):
        eng = _engine(probs, mode="mesolve")
        eng.set_path(force)
        st = eng.new_state()
        snaps = eng.solve(st, times).cpu().numpy()
        assert np.array_equal(snaps[-1], st.cpu().numpy())
        outs[force] = snaps
        launches = eng.stats()["n_launches"]
        assert (launches == 2) if not force else (launches > 50)  # split at the duplicated time
    assert np.max(np.abs(outs[False] - outs[True])) < 1e-13
    assert np.array_equal(outs[False][0], outs[False][1])
    tr = np.trace(outs[False][-1], axis1=1, axis2=2)
    assert np.max(np.abs(tr - 1.0)) < 1e-12
    if n <= 4:
        for b, p in enumerate(probs):
            ham = qp.build_hamiltonian(p)
            ref = qp.mesolve(ham, qp.all_ground_state(n, p["eigenbasis"]), times[[0, 1, 3]],
                             max_step=1e-3, **qp.TIGHT)
            assert np.max(np.abs(outs[False][0][b] - ref[1])) < AMP_TOL
            assert np.max(np.abs(outs[False][2][b] - ref[2])) < AMP_TOL


@pytest.mark.parametrize("case", ["dephasing", "all"])
@pytest.mark.parametrize("n", [2, 4, 6])
def test_persistent_density_matrix_kernel_global_real_drive(case, n):
    """The uniform-real-drive variant (global channel, phase 0; one trajectory of the
    batch has a badly prepared atom, which is masked out of the partner sums)."""
    from oracle import qutip_path as qp

    ops, paulis = ME_CASES[case]
    base = chain_problem(n, collapse_ops=ops)
    base["depolarizing_pauli_2ds"] = dict(paulis)
    bad = dict(base)
    bad["bad_atoms"] = np.array([i == 1 for i in range(n)])
    probs = [base, bad]
    times = np.array([0.0, 0.9, 2.0])
    outs = {}
    for force in (False, True):
        eng = _engine(probs, mode="mesolve")
        eng.set_path(force)
        st = eng.new_state()
        outs[force] = eng.solve(st, times).cpu().numpy()
        assert (eng.stats()["n_launches"] == 1) == (not force)
    assert np.max(np.abs(outs[False] - outs[True])) < 1e-12
    assert np.max(np.abs(outs[False][-1][0] - outs[False][-1][1])) > 1e-3
    if n <= 4:
        for b, p in enumerate(probs):
            ham = qp.build_hamiltonian(p)
            ref = qp.mesolve(ham, qp.all_ground_state(n, p["eigenbasis"]), times, max_step=1e-3, **qp.TIGHT)
            assert np.max(np.abs(outs[False][-1][b] - ref[-1])) < AMP_TOL


@pytest.mark.parametrize("mode,n", [("sesolve", 13), ("sesolve", 15), ("mesolve", 7)])
def test_single_launch_plan_with_different_problems_per_batch_entry(mode, n):
    """Three different local-addressing problems in one batch (own coefficients and own
    interaction diagonal per entry): the single-launch plan (partner tiles of the high
    bits read from global memory) against the multi-pass tiling, and - where the
    persistent kernel exists - against it."""
    ops = [(np.sqrt(0.1), "sigma_rr")] if mode == "mesolve" else None
    probs = [local_problem(n, seed=s, duration=31, collapse_ops=ops) for s in (3, 4, 5)]
    times = np.array([0.0, 0.004, 0.011])
    outs = {}
    for name, kw in (("single", {}), ("multi", {"no_single_pass": True})):
        eng = _engine(probs, mode=mode)
        eng.set_path(True, no_tile14=True, **kw)
        outs[name] = eng.solve(eng.new_state(), times).cpu().numpy()
        assert eng.stats()["passes"] == (1 if name == "single" else 2)
    assert np.max(np.abs(outs["single"] - outs["multi"])) < 1e-13
    if mode == "sesolve" and n <= 13:
        eng = _engine(probs, mode=mode)
        eng.set_path(False, no_split14=True)  # the persistent kernel (round 4: the default here is k_split_reg)
        assert np.max(np.abs(eng.solve(eng.new_state(), times).cpu().numpy() - outs["single"])) < 1e-12
    assert np.max(np.abs(outs["single"][-1][0] - outs["single"][-1][1])) > 1e-3  # really different problems
