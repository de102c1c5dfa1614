"""World-size-2 gloo tests (CPU) of the trajectory sharding: results must be
identical to the serial reference order for any world size."""
import os
import socket
from collections import Counter

import numpy as np
import pytest

from helpers import load_fixture
from test_host_logic import _chain12_inputs

from pulser_amd import NoiseModel, QutipEmulator
from pulser_amd.distributed import (flips_with, partition, predraw_sampling,
                                    run_ensemble, sample_with)
from pulser_amd.results import spam_flips


def test_partition_is_contiguous_and_balanced():
    reps = [900, 50, 30, 20, 10, 5, 5, 4]
    for world in (1, 2, 3, 4, 8):
        blocks = partition(reps, world)
        assert blocks[0][0] == 0 and blocks[-1][1] == len(reps)
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
    assert partition([1] * 1024, 8) == [(128 * r, 128 * (r + 1)) for r in range(8)]


def test_predrawn_sampling_equals_reference_call_sequence():
    rng = np.random.default_rng(0)
    w = rng.random(16)
    w /= w.sum()
    np.random.seed(4)
    from pulser_amd.results import multinomial

    idx = multinomial(200, w)
    c = Counter(np.binary_repr(i, 4) for i in idx)
    flipped = spam_flips(c, 0.03, 0.08)
    np.random.seed(4)
    (r, mat), = predraw_sampling([40], 1, 5, 4, True)[0]
    idx2 = sample_with(r, w)
    assert np.array_equal(idx, idx2)
    out = flips_with(idx2, 4, mat, 0.03, 0.08)
    assert Counter(np.binary_repr(i, 4) for i in out) == flipped


def _fake_states(problems, n_eval):
    """Deterministic stand-in for the HIP solver: a normalised state that
    depends on the trajectory's noise parameters only."""
    out = []
    for p in problems:
        seed = int(abs(p["samples"]["Local"]["ground-rydberg"][0]["det"][10]) * 1e6) % (2**31)
        seed += int(np.sum(p["bad_atoms"]))
        rng = np.random.default_rng(seed)
        st = rng.normal(size=(n_eval, 2 ** p["n_qudits"])) + 1j * rng.normal(size=(n_eval, 2 ** p["n_qudits"]))
        st /= np.linalg.norm(st, axis=1, keepdims=True)
        out.append(st)
    return np.stack(out)


def _make_emulator(n_traj=16):
    _, extra = load_fixture("cfg4_chain12_noise.npz")
    from pulser_amd import problem as P
    from pulser_amd.hamiltonian_data import single_global_channel

    coords = P.register_coords(P.square_rect(1, 5), float(extra["blockade_radius"]))
    s = {k: v[:400] for k, v in P.anneal_samples().items()}
    inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
    nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.1, p_false_pos=0.01,
                    p_false_neg=0.05, samples_per_run=7)
    np.random.seed(21)
    return QutipEmulator(inputs, noise_model=nm, n_trajectories=n_traj,
                         evaluation_times=[0.0, 0.2, 0.4])


def _serial_reference(emu):
    """The reference's loop (simulation.py:847-883) with the fake solver."""
    from pulser_amd.results import CoherentResults, QState, StateResult

    total = np.array([Counter() for _ in emu._eval_times_array])
    qids = tuple(emu.samples_obj.qubit_ids)
    me = {"epsilon": emu.noise_model.p_false_pos, "epsilon_prime": emu.noise_model.p_false_neg}
    for prob in emu._problems:
        st = _fake_states([prob], len(emu._eval_times_array))[0]
        res = CoherentResults(
            [StateResult(qids, "ground-rydberg", QState(s), True) for s in st], len(qids),
            "ground-rydberg", emu._eval_times_array, "ground-rydberg", me)
        total += np.array([res.sample_state(t, n_samples=emu.noise_model.samples_per_run * prob["reps"])
                           for t in emu._eval_times_array])
    return list(total)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from pulser_amd.distributed import init_process_group

    d = init_process_group("gloo")
    emu = _make_emulator()
    if rank != 0:  # other ranks must NOT rely on their own RNG state
        np.random.seed(999 + rank)
    n_eval = len(emu._eval_times_array)
    out = run_ensemble(emu, lambda probs: _fake_states(probs, n_eval), dist=d, batch=3, density_matrix=True)
    q.put((rank, out["histograms"], out["mean_occupations"], out["block"], out["density_matrices"]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ensemble_equals_serial_reference_world1_and_world2(world):
    import torch.multiprocessing as mp

    emu = _make_emulator()
    ref = _serial_reference(emu)  # consumes the global RNG exactly like the reference
    # world size 1 (same process, fresh emulator with the same seed)
    emu1 = _make_emulator()
    n_eval = len(emu1._eval_times_array)
    out1 = run_ensemble(emu1, lambda probs: _fake_states(probs, n_eval), dist=None, batch=4, density_matrix=True)
    # the ensemble density matrix = reps-weighted mean of |psi><psi| (aggregators.py:19-37)
    rho_ref = np.zeros_like(out1["density_matrices"])
    for prob in emu1._problems:
        st = _fake_states([prob], n_eval)[0]
        rho_ref += prob["reps"] * np.einsum("ti,tj->tij", st, st.conj())
    assert np.allclose(out1["density_matrices"], rho_ref / 16, atol=1e-14)
    assert np.allclose(np.trace(out1["density_matrices"], axis1=1, axis2=2), 1.0, atol=1e-13)
    assert out1["counters"] == ref
    assert out1["n_measures"] == 16 * 7
    # world size 2 over gloo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blocks = sorted(g[3] for g in got)  # contiguous shards covering every trajectory once
    assert blocks[0][0] == 0 and blocks[-1][1] == len(emu._problems)
    assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    for _, hist, occ, _, rho in got:
        assert np.array_equal(hist, out1["histograms"])
        assert np.allclose(occ, out1["mean_occupations"], atol=1e-14)
        assert np.allclose(rho, out1["density_matrices"], atol=1e-14)  # the all-reduced sum of |psi><psi|


def test_run_hands_validated_options_to_the_ensemble_without_a_second_validation():
    """QutipEmulator.run() validates its options and hands the dict to run_ensemble(): a second validation
    would take the filled-in DEFAULT max_step for a requested one and switch the multi-knot CF4 steps off
    for every sharded run (serial and sharded runs would then step differently)."""
    emu = _make_emulator()
    opts: dict = {}
    emu._validate_options(opts)
    assert "max_step" in opts and "max_step" not in emu._engine_kwargs(opts)
    n_eval = len(emu._eval_times_array)
    seen = []
    orig = emu._validate_options
    emu._validate_options = lambda o: (seen.append(dict(o)), orig(o))[1]
    run_ensemble(emu, lambda probs: _fake_states(probs, n_eval), dist=None, options=opts, options_validated=True)
    assert seen == [] and "max_step" not in emu._engine_kwargs(opts)
    run_ensemble(emu, lambda probs: _fake_states(probs, n_eval), dist=None, options={})  # direct callers: once
    assert len(seen) == 1 and "max_step" not in emu._engine_kwargs(seen[0] | {"max_step": 1e-3})
    asked = {"max_step": 0.002}
    emu._validate_options(asked)
    assert emu._engine_kwargs(asked)["max_step"] == 0.002


def _digest_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from pulser_amd.distributed import check_same_problem, enable_sharding, init_process_group

    d = init_process_group("gloo")
    emu = _make_emulator()
    assert emu._distributed() is None  # no opt-in, no sharding
    enable_sharding()
    assert emu._distributed() is not None
    check_same_problem(d, emu)  # identical on every rank: passes
    other = _make_emulator(n_traj=16 if rank == 0 else 8)  # ranks disagree
    try:
        check_same_problem(d, other)
        q.put((rank, "accepted"))
    except RuntimeError as exc:
        q.put((rank, str(exc)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_is_an_opt_in_and_refuses_ranks_that_run_different_jobs():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_digest_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, msg in got:
        assert msg.startswith("Sharded run refused: ranks [1]"), msg


def test_batched_cumulative_weights_are_bit_identical_to_the_per_state_path():
    """The block-wise weights / cumulative sums of the sharded fast path against ``StateResult._weights`` +
    ``np.cumsum`` state by state: the sampled indices must not depend on the batching."""
    from pulser_amd.distributed import cumulative_weights
    from pulser_amd.results import QState, StateResult

    rng = np.random.default_rng(0)
    n, D = 6, 64
    kets = rng.normal(size=(3, 5, D)) + 1j * rng.normal(size=(3, 5, D))
    kets /= np.linalg.norm(kets, axis=-1, keepdims=True) * (1 + 1e-7 * rng.normal(size=(3, 5, 1)))  # norms 1 +- 1e-7
    kets[0, :, :] = 0.0
    kets[0, :, -1] = 1.0  # the all-ground initial state
    diags = np.abs(rng.normal(size=(2, 4, D))) + 0j
    qids = tuple(f"q{i}" for i in range(n))
    for basis, matching in (("ground-rydberg", True), ("digital", True), ("ground-rydberg", False)):
        for arr, is_ket in ((kets, True), (diags, False)):
            cum = cumulative_weights(arr, is_ket, basis, matching)
            for a in range(arr.shape[0]):
                for b in range(arr.shape[1]):
                    st = QState(arr[a, b]) if is_ket else QState(np.diag(arr[a, b]))
                    ref = np.cumsum(StateResult(qids, basis, st, matching)._weights())
                    assert np.array_equal(cum[a, b], ref), (basis, matching, is_ket, a, b)
