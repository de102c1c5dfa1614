"""Trajectory sharding over the GPUs of one node (one process per GPU).

The reference runs noise trajectories in a serial Python loop
(pulser-simulation/pulser_simulation/simulation.py:850-861, 903-915).  Each
iteration is independent once its random parameters are drawn, so the
(trajectory, reps) list is split into contiguous blocks balanced by reps, one
block per rank, with NO data-path collective.  What keeps results identical to
the serial reference - and independent of the world size - is that every random
number is drawn on rank 0 from the global ``np.random`` stream in the
reference's order:

1. noise-trajectory parameters at construction (hamiltonian_data.py:782-911),
2. per (trajectory, evaluation time): ``rand(n)`` for the multinomial draw
   (math/multinomial.py:32) and, with SPAM measurement errors,
   ``uniform(size=(n, N))`` for the bit flips (simresults.py:556-558),

and broadcast.  The only collective on the result path is one all-reduce (sum)
of the per-rank accumulators: bitstring histograms int64[n_eval, 2^N]
(BAG_UNION of Counters == sum), reps-weighted occupation sums float64[n_eval, N+1]
and - on request - the trajectory sum of |psi><psi| (density_matrix_aggregator,
pulser_simulation/aggregators.py:19-37).  Backend: ``nccl`` (= RCCL over xGMI) on
GPUs, ``gloo`` in the CPU tests.
"""

from __future__ import annotations

import os
from collections import Counter
from typing import Any, Callable, Sequence

import numpy as np


# Sharding is an explicit opt-in (a process group may exist for unrelated reasons - a parameter sweep
# with one sequence per rank, data-parallel training around the emulator - and implicit collectives
# would deadlock it or sum unrelated histograms): ``enable_sharding()`` or PULSER_AMD_SHARD=1.
_SHARDING = {"enabled": os.environ.get("PULSER_AMD_SHARD", "0") not in ("", "0")}


def enable_sharding(enabled: bool = True) -> None:
    """Let noisy ``QutipEmulator.run()`` / ``QutipBackendV2.run()`` split their trajectories over the
    ranks of the initialised ``torch.distributed`` group.  EVERY rank must then make the same call on
    the same sequence, noise model and evaluation times (checked, see :func:`check_same_problem`)."""
    _SHARDING["enabled"] = bool(enabled)


def sharding_enabled() -> bool:
    return _SHARDING["enabled"]


def problem_digest(emulator: Any) -> str:
    """SHA-256 over what defines a run: the sampled sequence, register, noise model, evaluation times,
    trajectory count, sampling rate, solver."""
    import hashlib
    import pickle

    h = hashlib.sha256()

    def feed(x: Any) -> None:
        if isinstance(x, dict):
            for k in sorted(x):
                h.update(str(k).encode())
                feed(x[k])
        elif isinstance(x, (list, tuple)):
            h.update(b"[")
            for v in x:
                feed(v)
            h.update(b"]")
        elif isinstance(x, np.ndarray):
            h.update(str(x.dtype).encode() + str(x.shape).encode())
            h.update(np.ascontiguousarray(x).tobytes())
        else:
            h.update(repr(x).encode())

    feed(emulator.samples_obj.to_dict())
    feed(np.asarray(emulator._eval_times_array, dtype=np.float64))
    feed(repr(emulator.noise_model))
    feed((emulator.n_trajectories, float(emulator._sampling_rate), str(emulator.solver)))
    try:
        feed(np.asarray(emulator._initial_state))
    except Exception:  # pragma: no cover
        h.update(pickle.dumps(None))
    return h.hexdigest()


def check_same_problem(dist: Any, emulator: Any) -> None:
    """Every rank must be running the same emulation before any collective mixes their results."""
    mine = problem_digest(emulator)
    digests: list[Any] = [None] * dist.get_world_size()
    dist.all_gather_object(digests, mine)
    if any(d != digests[0] for d in digests):
        bad = [r for r, d in enumerate(digests) if d != digests[0]]
        raise RuntimeError(
            "Sharded run refused: ranks " + str(bad) + " hold a different sequence / noise model / "
            "evaluation times than rank 0. Sharding splits ONE emulation over the ranks; disable it "
            "(pulser_amd.distributed.enable_sharding(False)) when the ranks run different jobs.")


def cumulative_weights(states: np.ndarray, is_ket: bool, meas_basis: str, matching: bool) -> np.ndarray:
    """``np.cumsum(QutipResult._weights())`` of TWO-LEVEL states for a whole array [..., 2^N] of kets (or of
    density-matrix diagonals) at once.  Bit-identical to the per-state path: ``np.abs(psi) ** 2`` /
    ``np.abs(diag)`` are elementwise, the reversal is a view, and ``cumsum`` along the last axis is the same
    sequential left-to-right sum as the 1-D call (qutip_result.py:101-118, 158; multinomial.py:32-36)."""
    probs = np.abs(states) ** 2 if is_ket else np.abs(states)
    if matching:
        weights = probs[..., ::-1] if meas_basis == "ground-rydberg" else probs
    else:  # qutip_result.py:119-122
        weights = np.zeros(probs.shape)
        weights[..., 0] = 1.0
    w = weights / np.cumsum(weights, axis=-1)[..., -1:]
    return np.cumsum(w, axis=-1)


def replay_block_native(host: np.ndarray, is_ket: bool, meas_basis: str, matching: bool, n: int,
                        starts: np.ndarray, counts: np.ndarray, rnd_all: np.ndarray, mat_all: np.ndarray | None,
                        eps: float, eps_p: float, hist: np.ndarray, n_threads: int = 0) -> None:
    """``hist[ti] += bincount(flips(searchsorted(cumulative_weights(host[ti, j]), rnd)))`` for every (ti, j) of a block in
    one call of ``ryd_replay_samples`` (host threads, no GIL): ``host`` complex128[n_eval, B, 2^n] kets or diagonals,
    ``starts`` / ``counts`` int64[n_eval, B] the slices of ``rnd_all`` (and rows of ``mat_all``).  Same histograms as
    ``cumulative_weights`` + ``np.searchsorted`` + ``flips_with`` row by row (tests/test_host_logic.py)."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    n_eval, B, D = host.shape
    host = np.ascontiguousarray(host, dtype=np.complex128)
    st = np.ascontiguousarray(starts, dtype=np.int64).reshape(-1)
    ct = np.ascontiguousarray(counts, dtype=np.int64).reshape(-1)
    slot = np.repeat(np.arange(n_eval, dtype=np.int32), B)
    rnd = np.ascontiguousarray(rnd_all, dtype=np.float64)
    mat = None if mat_all is None else np.ascontiguousarray(mat_all, dtype=np.float64)
    if not (hist.flags.c_contiguous and hist.dtype == np.int64 and hist.shape == (n_eval, D)):
        raise ValueError("hist must be a C-contiguous int64[n_eval, 2^n] array")
    _lib.check(lib.ryd_replay_samples(
        host.ctypes.data, n_eval * B, D, n, int(is_ket), int(matching and meas_basis == "ground-rydberg"), int(matching),
        st.ctypes.data, ct.ctypes.data, slot.ctypes.data, n_eval, rnd.ctypes.data,
        None if mat is None else mat.ctypes.data, float(eps), float(eps_p), hist.ctypes.data, int(n_threads)))


class _DiagonalState:
    """What the sampling chain needs of a density matrix (qutip_result.py:101-118): its diagonal."""

    isket = False

    def __init__(self, diag: np.ndarray) -> None:
        self._diag = np.asarray(diag)
        self.shape = (self._diag.size, self._diag.size)

    def diag(self) -> np.ndarray:
        return self._diag

    def tr(self) -> complex:
        return complex(np.sum(self._diag))


def env_world() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str | None = None) -> Any:
    """Initialise ``torch.distributed`` (no-op for a single process)."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def partition(reps: Sequence[int], world: int) -> list[tuple[int, int]]:
    """Contiguous [start, stop) blocks of the trajectory list, balanced by the
    cumulative number of repetitions."""
    reps = np.asarray(reps, dtype=np.int64)
    total = int(reps.sum())
    cum = np.concatenate([[0], np.cumsum(reps)])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(max(i, bounds[-1]), len(reps)))
    bounds.append(len(reps))
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def sample_with(rnd: np.ndarray, weights: np.ndarray) -> np.ndarray:
    """``multinomial`` with pre-drawn uniforms (math/multinomial.py:32-36)."""
    return np.searchsorted(np.cumsum(weights), rnd)


def flips_with(indices: np.ndarray, n_qudits: int, rnd_matrix: np.ndarray | None,
               eps: float, eps_p: float) -> np.ndarray:
    """Measurement flips of simresults.py:537-568 with a pre-drawn matrix.

    The reference groups identical shots (Counter, first-occurrence order) and
    repeats each group; the random matrix rows are consumed in that order."""
    if rnd_matrix is None or (eps == 0.0 and eps_p == 0):
        return indices
    # the reference's Counter: distinct outcomes in order of first occurrence, with their multiplicities
    uniq, first, cnt = np.unique(np.asarray(indices, dtype=np.int64), return_index=True, return_counts=True)
    order = np.argsort(first, kind="stable")
    shots, counts = uniq[order], cnt[order]
    bits = (shots[:, None] >> (n_qudits - 1 - np.arange(n_qudits))[None, :]) & 1
    flip_probs = np.where(bits == 1, eps_p, eps)
    flips = rnd_matrix < np.repeat(flip_probs, counts, axis=0)
    new_bits = np.repeat(bits, counts, axis=0) ^ flips
    weights = 1 << (n_qudits - 1 - np.arange(n_qudits))
    return (new_bits * weights[None, :]).sum(axis=1)


def predraw_sampling(reps: Sequence[int], n_eval: int, samples_per_run: int, n_qudits: int,
                     with_meas_errors: bool) -> list[list[tuple[np.ndarray, np.ndarray | None]]]:
    """Rank 0: all sampling random numbers in the reference's order
    (trajectory-major, evaluation-time-minor; simulation.py:853-861)."""
    out = []
    for r in reps:
        n = samples_per_run * int(r)
        per_t = []
        for _ in range(n_eval):
            rnd = np.random.rand(n)
            mat = np.random.uniform(size=(n, n_qudits)) if with_meas_errors else None
            per_t.append((rnd, mat))
        out.append(per_t)
    return out


def _broadcast_array(dist: Any, arr: np.ndarray | None, shape: tuple[int, ...], src: int = 0) -> np.ndarray:
    """One contiguous float64 tensor broadcast (device tensor under nccl, host under gloo)."""
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = (torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(dev) if arr is not None
         else torch.empty(shape, dtype=torch.float64, device=dev))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


# pinned staging buffers of the sampling replay, kept between ensembles (allocating 67 MB of pinned host memory costs
# more than copying it); one buffer per (shape, dtype), at most two kept
_PINNED_CACHE: dict[tuple, Any] = {}


def _pinned_like(torch: Any, dev_t: Any) -> Any:
    key = (tuple(dev_t.shape), str(dev_t.dtype))
    buf = _PINNED_CACHE.get(key)
    if buf is None:
        while len(_PINNED_CACHE) >= 2:
            _PINNED_CACHE.pop(next(iter(_PINNED_CACHE)))
        buf = _PINNED_CACHE[key] = torch.empty(key[0], dtype=dev_t.dtype, pin_memory=True)
    return buf


def run_ensemble(
    emulator: Any,
    solve_fn: Callable[[list[dict[str, Any]]], np.ndarray] | None = None,
    dist: Any = None,
    batch: int = 128,
    mc_seed: int | None = None,
    options: dict[str, Any] | None = None,
    density_matrix: bool = False,
    options_validated: bool = False,
) -> dict[str, Any]:
    """Sharded equivalent of the stochastic branch of ``QutipEmulator.run``
    (simulation.py:847-883, 885-915); ``QutipEmulator.run`` calls it when
    ``torch.distributed`` is initialised with more than one rank.

    Rank 0 owns every random draw - the noise trajectories (redrawn on a repeated run, like
    simulation.py:892-902) and the sampling uniforms - and broadcasts them: the trajectory list as
    objects (a few KB each), the uniforms as two contiguous float64 tensors.  Every rank solves its
    contiguous block on its GPU and samples it; ONE all-reduce (sum) per accumulator closes the run:
    bitstring histograms int64[n_eval, 2^N], measured-bit occupation sums float64[n_eval, N + 1]
    and, with ``density_matrix``, the reps-weighted sum of |psi><psi| (aggregators.py:19-37) as
    float64[n_eval, D, D, 2].  ``solve_fn(problems)`` (tests) returns host states
    complex[len(problems), n_eval, dim] instead of running the HIP engine.  On the tuned two-level
    path the states never leave the device except for the kets the bit-exact sampling replay reads:
    occupations come from ``ryd_occupations``, the |psi><psi| sum from ``ryd_outer_accumulate_dim``
    and the result ``density_matrices`` is a CUDA tensor.
    """
    import time as _time

    import torch

    t_start = _time.perf_counter()
    tm: dict[str, float] = {"lower_wait_ms": 0.0, "solve_ms": 0.0, "post_ms": 0.0}  # where the wall time went (bench detail)
    rank, _, world = env_world()
    if dist is None:
        world, rank = 1, 0
    options = dict(options or {})
    # max_step / nsteps defaults, the SPAM + initial-state refusal - once: validating the dict run() already
    # validated would take the filled-in DEFAULT max_step for a requested one (and switch the multi-knot
    # steps off for every sharded run)
    if not options_validated:
        emulator._validate_options(options)
    if dist is not None:
        check_same_problem(dist, emulator)
    nm = emulator.noise_model
    times = emulator._eval_times_array
    n_eval = len(times)
    meas_err = "SPAM" in nm.noise_types and not (nm.p_false_pos == 0.0 and nm.p_false_neg == 0)
    # -- rank 0 owns every random draw ------------------------------------
    payload: list[Any] = [None]
    if rank == 0:
        emulator._refresh_trajectories_if_used()  # fresh draws on every further run (simulation.py:892-902)
        trajs = emulator._hamiltonian_data.noise_trajectories
        # quantum-jump seeds (used only when the solver is the Monte-Carlo one):
        # like qutip.mcsolve's, NOT from the global np.random stream
        mc_seeds = np.random.default_rng(mc_seed).integers(0, 2**64, size=len(trajs), dtype=np.uint64)
        payload = [(trajs, mc_seeds)]
    emulator._noise_trajectories_used = True
    if dist is not None:
        dist.broadcast_object_list(payload, src=0)
    trajs, mc_seeds = payload[0]
    hd = emulator._hamiltonian_data  # only rank 0's draws (`trajs`) are lowered below
    n = hd.n_qudits
    reps = np.array([t.reps for t in trajs], dtype=np.int64)
    shots = nm.samples_per_run * reps  # per (trajectory, evaluation time)
    offs = np.concatenate([[0], np.cumsum(np.repeat(shots, n_eval))])  # trajectory-major, time-minor
    total = int(offs[-1])
    lo, hi = partition(reps, world)[rank]
    # the factored lowering of this rank's FIRST block starts now, on its worker thread, and runs while rank 0 draws the
    # sampling uniforms below (the legacy generator releases the GIL while it fills an array); round 6: it used to start
    # after the draws and the first solve waited 10 - 15 ms for it
    lower_pool = None
    lowered: dict[int, Any] = {}
    if solve_fn is None and lo < hi and emulator._fast_path_ok(emulator._current_problem):
        from concurrent.futures import ThreadPoolExecutor as _TPE0

        lower_pool = _TPE0(max_workers=1)
        lowered[lo] = lower_pool.submit(hd.device_tables, [trajs[i] for i in range(lo, min(hi, lo + batch))],
                                        emulator._sampling_rate)
    rnd_all = mat_all = None
    if rank == 0:  # the reference's call sequence: rand(n), then uniform(size=(n, N)) (simulation.py:853-861)
        rnd_all = np.empty(total)
        mat_all = np.empty((total, n)) if meas_err else None
        for k in range(len(trajs) * n_eval):
            a, b = offs[k], offs[k + 1]
            rnd_all[a:b] = np.random.rand(b - a)
            if meas_err:
                mat_all[a:b] = np.random.uniform(size=(b - a, n))
    if dist is not None:
        rnd_all = _broadcast_array(dist, rnd_all, (total,))
        if meas_err:
            mat_all = _broadcast_array(dist, mat_all, (total, n))
    tm["draws_done_ms"] = (_time.perf_counter() - t_start) * 1e3
    n_traj = int(reps.sum())
    hist = np.zeros((n_eval, 2**n), dtype=np.int64)
    occ_sum = np.zeros((n_eval, n + 1), dtype=np.float64)
    rho_sum = None  # host accumulator (``solve_fn`` given / general path): complex128[n_eval, D, D]
    rho_dev = None  # device accumulator of the tuned path: the MEAN (weights reps / n_traj)
    default_solver = solve_fn is None
    fast = default_solver and emulator._fast_path_ok(emulator._current_problem)
    if default_solver:
        def solve_fn(problems: list[dict[str, Any]]) -> np.ndarray:
            res = emulator._solve_batch(problems, False, options)
            return np.stack([[np.asarray(s) for s in r.states] for r in res])
    else:
        mc_seeds = None
    from .results import QState, StateResult

    qids = tuple(emulator.samples_obj.qubit_ids)
    matching = emulator._meas_basis in emulator.basis_name  # (a property that walks the channels)
    bit_of = ((np.arange(2**n)[:, None] >> (n - 1 - np.arange(n))[None, :]) & 1).astype(np.float64)

    def sample(i: int, ti: int, st: Any) -> np.ndarray:
        """Bit-exact replay of the reference's sampling of trajectory i at evaluation time ti."""
        w = StateResult(qids, emulator._meas_basis, st, matching)._weights()
        k = i * n_eval + ti
        ind = sample_with(rnd_all[offs[k]:offs[k + 1]], w)
        ind = flips_with(ind, n, mat_all[offs[k]:offs[k + 1]] if meas_err else None,
                         nm.p_false_pos, nm.p_false_neg)
        hist[ti] += np.bincount(ind, minlength=2**n)
        return w

    # the C replay (ryd_replay_samples: the same arithmetic on host threads, outside the GIL); PULSER_AMD_NUMPY_REPLAY=1
    # keeps the NumPy replay (A/B, and what the CPU tests compare it with)
    native_replay = fast and not os.environ.get("PULSER_AMD_NUMPY_REPLAY")
    pinned: Any = None       # the replay worker's pinned staging buffer and copy stream (one worker: used block after block)
    copy_stream: Any = None
    pool = None
    pending: list[Any] = []
    try:
        for start in range(lo, hi, batch):
            block = list(range(start, min(hi, start + batch)))
            if mc_seeds is not None:  # a trajectory's jumps depend on its seed only, not on the sharding
                emulator._mc_seed_override = mc_seeds[block]
            try:
                if fast:
                    # factored lowering (shared spline tables + per-(trajectory, atom) scales); the states
                    # stay on the device: occupations / norms are reduced there (ryd_occupations), the
                    # reps-weighted sum of |psi><psi| is formed there (ryd_outer_accumulate_dim, fp64 matrix
                    # cores) and only the kets (or, for density matrices, their diagonals) of the
                    # bit-exact sampling replay cross PCIe - no D x D array exists on the host
                    import torch as _t

                    from .engine import accumulate, outer_accumulate

                    # the factored lowering of the NEXT block runs on a worker thread while this block is solved
                    if lower_pool is None:
                        from concurrent.futures import ThreadPoolExecutor as _TPE

                        lower_pool = _TPE(max_workers=1)
                    if start not in lowered:
                        lowered[start] = lower_pool.submit(hd.device_tables, [trajs[i] for i in block], emulator._sampling_rate)
                    nxt = start + batch
                    if nxt < hi and nxt not in lowered:
                        lowered[nxt] = lower_pool.submit(
                            hd.device_tables, [trajs[i] for i in range(nxt, min(hi, nxt + batch))], emulator._sampling_rate)
                    t_a = _time.perf_counter()
                    tables = lowered.pop(start).result()
                    t_b = _time.perf_counter()
                    first_dev, snaps_dev, occ = emulator._solve_batch([], False, options, tables=tables, raw=True)
                    t_c = _time.perf_counter()
                    tm["lower_wait_ms"] += (t_b - t_a) * 1e3
                    tm["solve_ms"] += (t_c - t_b) * 1e3
                    is_ket = first_dev.dim() == 2
                    rb = reps[block].astype(np.float64)
                    norm = occ[..., n]  # [n_eval, B]
                    if not matching:  # weights = delta_0 (qutip_result.py:119-122): every measured bit is 0
                        bits = np.zeros_like(occ[..., :n])
                    elif emulator._meas_basis == "ground-rydberg":  # bit 1 <=> |r> = local index 0
                        bits = occ[..., :n] / norm[..., None]
                    else:  # digital: bit 1 <=> |h> = local index 1
                        bits = (norm[..., None] - occ[..., :n]) / norm[..., None]
                    occ_sum[:, :n] += np.einsum("b,tbk->tk", rb, bits)
                    occ_sum[:, n] += norm @ rb
                    if density_matrix:
                        D = int(first_dev.shape[1])
                        if rho_dev is None:
                            rho_dev = _t.zeros((n_eval, D, D), dtype=_t.complex128, device=first_dev.device)
                        for ti in range(n_eval):
                            x = first_dev if ti == 0 else snaps_dev[ti - 1]
                            if is_ket:
                                outer_accumulate(x, rho_dev[ti], rb / n_traj)
                            else:
                                for j in range(len(block)):
                                    accumulate(x[j], rho_dev[ti], float(rb[j]) / n_traj)
                    if is_ket:
                        dev_all = _t.cat([first_dev[None], snaps_dev])  # [n_eval, B, D]
                    else:
                        dev_all = _t.cat([_t.diagonal(first_dev, dim1=-2, dim2=-1)[None],
                                          _t.diagonal(snaps_dev, dim1=-2, dim2=-1)]).contiguous()
                    ready = _t.cuda.Event()
                    ready.record(_t.cuda.current_stream(dev_all.device))
                    del first_dev, snaps_dev
                    # the reference's weights and cumulative sums (qutip_result.py:101-158, multinomial.py:32-36) for
                    # the whole block at once: the same elementwise operations and the same sequential row sums
                    # ... on a worker thread, so that the replay of this block overlaps the solve of the next one
                    # (the GPU work is asynchronous C calls; the histogram is only read after the last block)
                    def replay(dev_all: Any = dev_all, ready: Any = ready, block: list[int] = block, is_ket: bool = is_ket) -> None:
                        # device -> pinned host memory on the worker's own stream (round 6: `.cpu()` on the main thread
                        # was a pageable copy of 33 MB per block, 7 ms during which no solve could be queued)
                        nonlocal pinned, copy_stream
                        if copy_stream is None:
                            copy_stream = _t.cuda.Stream(device=dev_all.device)
                        pinned = _pinned_like(_t, dev_all)
                        with _t.cuda.stream(copy_stream):
                            copy_stream.wait_event(ready)
                            pinned.copy_(dev_all, non_blocking=True)
                        copy_stream.synchronize()
                        del dev_all
                        host = pinned.numpy()
                        if native_replay:
                            ks = np.asarray(block, dtype=np.int64)[None, :] * n_eval + np.arange(n_eval, dtype=np.int64)[:, None]
                            replay_block_native(host, is_ket, emulator._meas_basis, matching, n, offs[ks], offs[ks + 1] - offs[ks],
                                                rnd_all, mat_all if meas_err else None, nm.p_false_pos, nm.p_false_neg, hist)
                            return
                        cum = cumulative_weights(host, is_ket, emulator._meas_basis, matching)
                        for j, i in enumerate(block):
                            for ti in range(n_eval):
                                k = i * n_eval + ti
                                ind = np.searchsorted(cum[ti, j], rnd_all[offs[k]:offs[k + 1]])
                                ind = flips_with(ind, n, mat_all[offs[k]:offs[k + 1]] if meas_err else None,
                                                 nm.p_false_pos, nm.p_false_neg)
                                hist[ti] += np.bincount(ind, minlength=2**n)

                    if pool is None:
                        from concurrent.futures import ThreadPoolExecutor

                        pool = ThreadPoolExecutor(max_workers=1)  # one worker: the replays run in block order
                    pending.append(pool.submit(replay))
                    tm["post_ms"] += (_time.perf_counter() - t_c) * 1e3
                    continue
                states = solve_fn([hd.problem(trajs[i], emulator._sampling_rate) for i in block])
            finally:
                emulator._mc_seed_override = None
            for j, i in enumerate(block):
                for ti in range(n_eval):
                    st = QState(states[j][ti])
                    w = sample(i, ti, st)
                    # measured-bit occupations from the 2^N weights: valid for every basis (2-, 3-, 4-level)
                    occ_sum[ti, :n] += reps[i] * (w @ bit_of)
                    occ_sum[ti, n] += reps[i] * (float(np.vdot(st, st).real) if st.isket else float(st.tr().real))
                    if density_matrix:
                        # explicit-term general path (multi-level / XY registers of a few atoms) and the
                        # host ``solve_fn`` of the CPU tests
                        a = np.asarray(st)
                        r1 = (a @ a.conj().T) if st.isket else a
                        if rho_sum is None:
                            rho_sum = np.zeros((n_eval,) + r1.shape, dtype=np.complex128)
                        rho_sum[ti] += reps[i] * r1
        t_a = _time.perf_counter()
        for fut in pending:
            fut.result()  # (re-raises what a replay raised)
        tm["replay_tail_ms"] = (_time.perf_counter() - t_a) * 1e3
    finally:
        # on an error too: nothing may keep running behind the caller's back (a prefetched lowering holds device
        # tables, a queued replay would go on adding to `hist`)
        for fut in list(lowered.values()) + pending:
            fut.cancel()
        if pool is not None:
            pool.shutdown(wait=True)
        if lower_pool is not None:
            lower_pool.shutdown(wait=True)
    # -- the one collective per accumulator: sum over ranks -----------------
    on_device = fast and density_matrix
    if density_matrix and not on_device and rho_sum is None:  # an empty shard still takes part in the all-reduce
        d = len(hd.eigenbasis) ** n
        rho_sum = np.zeros((n_eval, d, d), dtype=np.complex128)
    if on_device and rho_dev is None:
        import torch as _t

        d = len(hd.eigenbasis) ** n
        rho_dev = _t.zeros((n_eval, d, d), dtype=_t.complex128, device=_t.device("cuda", _t.cuda.current_device()))
    if dist is not None:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        th = torch.from_numpy(hist).to(dev)
        to = torch.from_numpy(occ_sum).to(dev)
        dist.all_reduce(th)
        dist.all_reduce(to)
        hist, occ_sum = th.cpu().numpy(), to.cpu().numpy()
        if on_device:  # the device tensor goes into the all-reduce as it is (RCCL); gloo needs a host hop
            tr = torch.view_as_real(rho_dev)
            if dev == "cpu":
                tr = tr.cpu()
            dist.all_reduce(tr)
            rho_dev = torch.view_as_complex(tr.to(rho_dev.device))
        elif density_matrix:
            tr = torch.view_as_real(torch.from_numpy(rho_sum)).contiguous().to(dev)  # float64[..., 2]
            dist.all_reduce(tr)
            rho_sum = torch.view_as_complex(tr.cpu().contiguous()).numpy()
    counters = [
        Counter({np.binary_repr(i, n): int(c) for i, c in enumerate(h) if c})
        for h in hist
    ]
    out = {
        "histograms": hist,
        "counters": counters,
        "mean_occupations": occ_sum[:, :n] / n_traj,
        "mean_norm": occ_sum[:, n] / n_traj,
        "n_measures": n_traj * nm.samples_per_run,
        "block": (lo, hi),
    }
    tm["total_ms"] = (_time.perf_counter() - t_start) * 1e3
    out["timings"] = tm
    if density_matrix:
        # tuned path: a CUDA tensor complex128[n_eval, D, D] (it never existed on the host; ``.cpu()`` on
        # request); host solvers: a NumPy array
        out["density_matrices"] = rho_dev if on_device else rho_sum / n_traj
    return out
