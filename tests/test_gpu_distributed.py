"""-m gpu: ``QutipEmulator.run()`` shards its noise trajectories when ``torch.distributed`` is
initialised (one process per GPU).  Here two ranks share the one GPU of the test box over gloo -
the RCCL path differs only in the backend of the one all-reduce."""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _emulator(seed):
    from pulser_amd import NoiseModel, QutipEmulator, problem as P
    from pulser_amd.hamiltonian_data import single_global_channel

    coords = P.register_coords(P.square_rect(1, 6), 8.0)
    s = {k: v[:600] for k, v in P.anneal_samples().items()}
    inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
    nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.05, p_false_pos=0.01,
                    p_false_neg=0.05, samples_per_run=9)
    np.random.seed(seed)
    return QutipEmulator(inputs, noise_model=nm, n_trajectories=24, evaluation_times=[0.0, 0.3, 0.6])


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import warnings

    import torch
    import torch.distributed as dist

    from pulser_amd.distributed import enable_sharding

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    emu = _emulator(5 if rank == 0 else 1234 + rank)  # only rank 0's random stream may matter
    assert emu._distributed() is None  # a process group alone does not shard anything (explicit opt-in)
    enable_sharding()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        res = emu.run()
    q.put((rank, [dict(r.bitstring_counts) for r in res], res.n_measures if hasattr(res, "n_measures") else None))
    dist.barrier()
    dist.destroy_process_group()


def test_run_shards_trajectories_over_ranks_and_matches_the_serial_run():
    import warnings

    import torch.multiprocessing as mp

    emu = _emulator(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        serial = emu.run()
    ref = [dict(r.bitstring_counts) for r in serial]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, counters, _ in got:
        assert counters == ref, rank  # bit-identical Counters on every rank, for any world size


def _v2_run(seed):
    from pulser_amd import NoiseModel, QutipBackendV2, QutipConfig, problem as P
    from pulser_amd.backend import BitStrings, Occupation
    from pulser_amd.hamiltonian_data import single_global_channel

    coords = P.register_coords(P.square_rect(1, 5), 8.0)
    s = {k: v[:500] for k, v in P.anneal_samples().items()}
    inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
    nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.05, p_false_pos=0.02, p_false_neg=0.06)
    obs = [BitStrings(evaluation_times=[0.5, 1.0], num_shots=40), Occupation(evaluation_times=[1.0])]
    cfg = QutipConfig(observables=obs, noise_model=nm, n_trajectories=12)
    np.random.seed(seed)
    res = QutipBackendV2(inputs, config=cfg).run()
    return (dict(res.get_result(obs[0], 0.5)), dict(res.get_result(obs[0], 1.0)),
            [float(v) for v in res.get_result(obs[1], 1.0)])


def _v2_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from pulser_amd.distributed import enable_sharding

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    enable_sharding()
    out = _v2_run(9 if rank == 0 else 777 + rank)  # only rank 0's stream may matter
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_backend_v2_run_shards_trajectories_and_matches_the_serial_run():
    """QutipBackendV2.run() under torch.distributed: ranks replay rank 0's random stream, observe
    only their own trajectories (skipping the BitStrings draws of the others) and gather the
    per-trajectory Results before aggregation (qutip_backend.py:266-325)."""
    import torch.multiprocessing as mp

    ref = _v2_run(9)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_v2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in got:
        assert out[0] == ref[0] and out[1] == ref[1], rank  # bit-identical bag-union Counters
        assert np.allclose(out[2], ref[2], atol=1e-12)      # mean occupations


def test_ensemble_density_matrix_and_occupations_stay_on_the_device_12_atoms():
    """SURVEY row A15 in the product path (aggregators.py:20-37, qutip_backend.py:322-325): at 12 atoms
    the trajectory mean of |psi><psi| is a 268 MB matrix - it is formed by ``ryd_outer_accumulate_dim`` on
    the device and never exists on the host; occupations come from ``ryd_occupations``."""
    import tracemalloc

    import torch

    from pulser_amd import NoiseModel, QutipEmulator, problem as P
    from pulser_amd.distributed import run_ensemble
    from pulser_amd.hamiltonian_data import single_global_channel

    n = 12
    coords = P.register_coords(P.square_rect(1, n), 8.692)
    s = {k: v[:300] for k, v in P.anneal_samples().items()}
    inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
    nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.005, p_false_pos=0.01, p_false_neg=0.05)
    np.random.seed(3)
    emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=24, evaluation_times=[0.0, 0.3])
    D = 2**n
    tracemalloc.start()
    out = run_ensemble(emu, dist=None, density_matrix=True)
    _, peak = tracemalloc.get_traced_memory()
    tracemalloc.stop()
    assert peak < 16 * D * D // 2, f"host peak {peak / 2**20:.0f} MiB: a D x D array was formed on the host"
    rho = out["density_matrices"]
    assert isinstance(rho, torch.Tensor) and rho.is_cuda and tuple(rho.shape) == (len(emu._eval_times_array), D, D)
    # consistency with the independently reduced occupations / norms and with the initial state
    bit_r = 1 - ((torch.arange(D, device=rho.device)[:, None] >> (n - 1 - torch.arange(n, device=rho.device))[None, :]) & 1)
    for ti in range(rho.shape[0]):
        diag = torch.diagonal(rho[ti]).real
        assert abs(float(diag.sum()) - out["mean_norm"][ti]) < 1e-12
        occ = (diag[:, None] * bit_r).sum(0).cpu().numpy()
        assert np.allclose(occ, out["mean_occupations"][ti], atol=1e-8)  # (per-trajectory norms are 1 +- 1e-9)
        assert float((rho[ti] - rho[ti].conj().T).abs().max()) < 1e-15
    assert abs(complex(rho[0, D - 1, D - 1]) - 1.0) < 1e-14  # t = 0: every trajectory starts in |g...g>
    assert out["mean_occupations"][-1].max() > 1e-3  # something happened


def test_v2_density_matrix_aggregation_runs_on_the_device():
    """``Results.aggregate`` with the StateResult observable: the mean of |psi><psi| is formed on the
    device (no host outer products) and equals the NumPy mean."""
    from unittest import mock

    from pulser_amd.backend import RydState, density_matrix_aggregator

    rng = np.random.default_rng(0)
    for d, n, eig in ((2, 7, ("r", "g")), (3, 4, ("r", "g", "h")), (4, 3, ("r", "g", "h", "x"))):
        D = d**n
        kets = rng.normal(size=(9, D)) + 1j * rng.normal(size=(9, D))
        kets /= np.linalg.norm(kets, axis=1)[:, None]
        ref = np.einsum("ti,tj->ij", kets, kets.conj()) / 9
        states = [RydState(k, eigenstates=eig) for k in kets]
        with mock.patch("numpy.outer", side_effect=AssertionError("host outer product")):
            got = density_matrix_aggregator(states)
        assert np.max(np.abs(np.asarray(got.to_qobj()) - ref)) < 1e-14
        # mixing kets and density matrices (aggregators.py:29-35)
        mixed = density_matrix_aggregator([states[0], RydState(ref, eigenstates=eig)])
        assert np.max(np.abs(np.asarray(mixed.to_qobj()) - 0.5 * (np.outer(kets[0], kets[0].conj()) + ref))) < 1e-14
