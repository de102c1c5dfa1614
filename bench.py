#!/usr/bin/env python
"""bench.py - sim-us/s of the emulation hot path on MI355X (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Primary workload (BASELINE.json configs[1]): 12-atom chain at the blockade
radius, analog Ising anneal (T = 3100 ns), sesolve fp64.  One "step" = one full
pass of the hot path (all 3.1 us) over a batch of `--batch` independent
sequences per GPU (default 256 = one per CU; cfg4 uses 128 per GPU), inputs
(spline tables, interaction diagonal, initial states) resident in HBM before the
timed region.  value = (GPUs x batch x 3.1 us) / seconds-per-step.  Multi-GPU:
sequences shard across ranks with no data-path collective; the only RCCL call
is the all-reduce of the ensemble occupation sums at the end of each step.

The JSON line also carries
  roofline      - the dominant kernel of the primary workload, timed live with
                  HIP events on the launch stream (librydemu's own event pairs);
  cpu_baseline  - the CPU oracle (SciPy restatement of the QuTiP path, the
                  reference's algorithm) on one host core, rank 0, N = 1 only;
  also          - secondary workloads (cfg3 14-atom Lindblad slice, cfg5 20-atom
                  sesolve slice, single-sequence latency) with their own rooflines.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
T_SEQ_US = 3.1


def blockade_radius() -> float:
    from pulser_amd import problem as P

    return (P.C6_LEVEL70 / (4 * 2 * np.pi / 2)) ** (1 / 6)


def chain_problem(n: int, collapse_ops=None):
    from pulser_amd import problem as P

    coords = P.register_coords(P.square_rect(1, n), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=collapse_ops)


def tri_problem(rows: int, cols: int, collapse_ops=None):
    from pulser_amd import problem as P

    coords = P.register_coords(P.triangular_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=collapse_ops)


def rect_problem(rows: int, cols: int):
    from pulser_amd import problem as P

    coords = P.register_coords(P.square_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples())


def timed_run(eng, state_fn, t0, t1, steps, warmup, dist=None, torch=None):
    """W warm-up + K timed passes [t0, t1]; returns (sec/step, stats, kernel ms, launches)."""
    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        st = state_fn()
        eng.evolve(st, t0, t1)
    barrier()
    eng.reset_stats()
    eng.set_kernel_timing(True)
    states = [state_fn() for _ in range(steps)]
    barrier()
    tic = time.perf_counter()
    occ = None
    def all_reduce(t, op=None):
        op = op if op is not None else dist.ReduceOp.SUM
        if dist.get_backend() == "gloo":  # RYD_BENCH_BACKEND=gloo: 1-GPU check of the N > 1 path
            c = t.cpu()
            dist.all_reduce(c, op=op)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op)

    for st in states:
        eng.evolve(st, t0, t1)
        occ = eng.occupations(st).sum(dim=0)
        if dist is not None:
            all_reduce(occ)  # ensemble sum over ranks (RCCL over xGMI)
    barrier()
    dt = time.perf_counter() - tic
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_reduce(tmax, dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kms, kl = eng.kernel_timing()
    eng.set_kernel_timing(False)
    return dt / steps, eng.stats(), kms, kl, occ


def measured_traffic(key):
    """HBM bytes per launch from the separate rocprofv3 --pmc passes of this
    round (tools/profile.sh -> profiles/*_traffic.json); None if not measured."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if key in d:
            return {"bytes_per_launch": d[key]["total_bytes_per_launch"],
                    "source": os.path.relpath(path, ROOT),
                    "method": "rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, separate passes"}
    return None


def roofline(nb, batch, stats, kernel_ms, launches, kernel_name, traffic_key=None):
    """Algorithmic bytes = 32 B x 2^nb per generator application."""
    apps = stats["n_applications"]
    bytes_total = 32.0 * (2.0**nb) * batch * apps
    sec = kernel_ms * 1e-3
    achieved = bytes_total / sec if sec > 0 else 0.0
    return {
        "bound": "hbm",
        "kernel": kernel_name,
        "achieved": achieved / 1e9,
        "peak": HBM_PEAK / 1e9,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK,
        "traffic": measured_traffic(traffic_key) if traffic_key else None,
        "launches": launches,
        "avg_launch_ms": kernel_ms / max(launches, 1),
        "applications": apps,
        "algorithmic_bytes_per_launch": bytes_total / max(launches, 1),
    }


def cpu_baseline(n: int):
    """QuTiP-path restatement (oracle) on ONE host core: full 3.1 us, N atoms."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle import qutip_path as qp

    prob = chain_problem(n)
    ham = qp.build_hamiltonian(prob)
    psi0 = qp.all_ground_state(n, prob["eigenbasis"])
    s = prob["samples"]["Global"]["ground-rydberg"]
    opts = qp.default_options([(s["amp"], s["det"])], 3100)
    counter = [0]
    tic = time.perf_counter()
    reps = 0
    while True:
        qp.sesolve(ham, psi0, np.array([0.0, T_SEQ_US]), counter=counter, **opts)
        reps += 1
        if time.perf_counter() - tic > 10.0 or reps >= 5:
            break
    dt = (time.perf_counter() - tic) / reps
    return {
        "value": T_SEQ_US / dt,
        "unit": "sim-us/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} x one {n}-atom sequence (3.1 us), SciPy CSR terms + not-a-knot spline + "
                  f"zvode Adams at QuTiP defaults (atol 1e-8, rtol 1e-6, max_step 1 ns): "
                  f"{counter[0] // reps} RHS evaluations per run, {dt:.2f} s per run",
        "host_cpu_count": os.cpu_count(),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="independent sequences per GPU")
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--slice-ns", type=int, default=2, help="cfg3/cfg5: simulated ns per step")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RYD_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:  # test hook: every rank on the one GPU of the box, gloo for the reductions
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist_mod.init_process_group(backend=backend)
        dist = dist_mod
    else:
        torch.cuda.set_device(0)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")

    from pulser_amd.engine import Engine

    n_gpus = max(world, 1)
    out = {}
    common = {"n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "f64", "data": "synthetic", "unit": "sim-us/s"}
    if args.workload == "cfg2":
        n, B = 12, args.batch
        eng = Engine.from_problems([chain_problem(n)] * B, mode="sesolve")
        sec, stats, kms, kl, occ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, args.steps,
                                             args.warmup, dist, torch)
        value = n_gpus * B * T_SEQ_US / sec
        out = {
            "metric": "sim-us/sec, 12-atom Rydberg anneal sequence, sesolve fp64 (aggregate over sequences)",
            "value": value,
            **common,
            "ms_per_step": sec * 1e3,
            "config": {
                "workload": "BASELINE configs[1]: 12-atom chain at the blockade radius, analog Ising "
                            "anneal 3100 ns, sesolve complex128; batch of independent sequences per GPU",
                "n_atoms": n,
                "sequences_per_gpu": B,
                "sim_us_per_sequence": T_SEQ_US,
                "integrator": "CF4 Magnus + Taylor(Horner), tol 1e-12/exponential",
                "taylor_order": stats["last_order"],
                "generator_applications_per_sequence": stats["n_applications"] // max(args.steps, 1),
                "parallelism": f"dp{n_gpus} (independent sequences, all-reduce of ensemble sums only)",
            },
            "roofline": roofline(n, B, stats, kms, kl, "k_traj<12,1024,1> (persistent, LDS-resident)",
                                 "cfg2:k_traj"),
        }
        out["roofline"]["note"] = (
            "state vectors stay in LDS/registers for the whole sequence, so the algorithmic 32 B/amp/"
            "application never reaches HBM; frac > 1 is on-chip reuse, not an HBM measurement - the "
            "kernel's real limit is the fp64 vector pipe, see 'compute'"
        )
        # What actually bounds the persistent kernel: fp64 VALU issue (profiles/r01_ktraj_counters.md).
        # Algorithmic flops per amplitude per application with a real global drive: diagonal 2,
        # N partner additions 2N, common coupling 2, Horner update 4 (FMA = 2 flops).
        flops_amp = 2 + 2 * n + 2 + 4
        tflops = flops_amp * (2.0**n) * B * stats["n_applications"] / (kms * 1e-3) / 1e12
        out["roofline"]["compute"] = {
            "bound": "valu_f64", "flops_per_amplitude_per_application": flops_amp,
            "achieved": tflops, "peak": 78.6, "unit": "TFLOP/s", "frac": tflops / 78.6,
            "note": "algorithmic flops only (address arithmetic, LDS traffic and barriers excluded); "
                    "PMC: ~90 % of the SIMD issue slots busy, 120 of 195 VALU instructions per wave "
                    "and stage are fp64 arithmetic",
        }
        eng.close()
    elif args.workload == "cfg4":
        # BASELINE configs[3]: 1024 noise trajectories of the 12-atom sequence, END TO END through
        # the emulator front-end (noise draws on rank 0, factored lowering, solve, reference-order
        # sampling), sharded over the ranks with one all-reduce of the histograms (strong scaling)
        from pulser_amd import NoiseModel, QutipEmulator, problem as P
        from pulser_amd.distributed import run_ensemble
        from pulser_amd.hamiltonian_data import single_global_channel

        coords = P.register_coords(P.square_rect(1, 12), blockade_radius())
        smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
        inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
        nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.005,
                        p_false_pos=0.01, p_false_neg=0.05)
        n_traj = 1024

        def one_pass(seed):
            np.random.seed(seed)
            emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=n_traj, evaluation_times="Minimal")
            return run_ensemble(emu, dist=dist, batch=256)

        for w in range(args.warmup):
            one_pass(100 + w)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        tic = time.perf_counter()
        for k in range(args.steps):
            res = one_pass(k)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        sec = (time.perf_counter() - tic) / args.steps
        if dist is not None:
            tmax = torch.tensor([sec], dtype=torch.float64)
            if dist.get_backend() != "gloo":
                tmax = tmax.cuda()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            sec = float(tmax.item())
        out = {"metric": "noise trajectories/s, 12-atom anneal sequence, end to end (draws, lowering, sesolve, sampling)",
               "value": n_traj / sec, **common, "unit": "trajectories/s", "scaling": "strong",
               "ms_per_step": sec * 1e3,
               "config": {"workload": "BASELINE configs[3]: 12-atom register, 1024 noise trajectories "
                                      "(doppler + amplitude + SPAM), sharded over the ranks, one all-reduce "
                                      "of the bitstring histograms", "n_atoms": 12, "n_trajectories": n_traj,
                          "sim_us_per_s": n_traj * T_SEQ_US / sec, "n_measures": int(res["n_measures"]),
                          "parallelism": f"dp{n_gpus} over trajectories"},
               "roofline": None}
        args.no_extras = True
        args.no_cpu = True
    elif args.workload in ("cfg3", "cfg5"):
        # HBM-streaming workloads as the primary line (used for the rocprofv3 passes)
        if args.workload == "cfg3":
            ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr")]
            eng = Engine.from_problems([tri_problem(2, 7, ops)], mode="mesolve")
            t0, t1, nb, kname = 1.0, 1.0 + 1e-3 * args.slice_ns, 28, "k_apply14<mesolve> + k_symm (Hermitian path: 2^14 register-tile row pass + tile-pair symmetrisation)"
            wl = f"BASELINE configs[2]: 14-atom triangular register, dephasing mesolve (rho = 4.29 GB), {args.slice_ns} ns slice at t = 1 us"
        else:
            eng = Engine.from_problems([rect_problem(4, 5)], mode="sesolve")
            t0, t1, nb, kname = 1.0, 1.0 + 1e-3 * args.slice_ns, 20, "k_apply<sesolve> (single-launch plan: 2^12 LDS tiles + 8 partner tiles through the Infinity Cache)"
            wl = f"BASELINE configs[4]: 20-atom 4x5 register, sesolve, {args.slice_ns} ns slice at t = 1 us"
        sec, stats, kms, kl, occ = timed_run(eng, eng.new_state, t0, t1, args.steps, args.warmup, dist, torch)
        out = {"metric": "sim-us/sec", "value": n_gpus * (t1 - t0) / sec, **common,
               "ms_per_step": sec * 1e3,
               "config": {"workload": wl, "passes_per_application": stats["passes"],
                          "taylor_order": stats["last_order"],
                          "parallelism": f"replicas x{n_gpus} (a single state does not shard)"},
               "roofline": roofline(nb, 1, stats, kms, kl, kname,
                                    "cfg3:k_apply" if args.workload == "cfg3" else None)}
        eng.close()
        args.no_extras = True
        args.no_cpu = True
    else:
        raise SystemExit(f"unknown workload {args.workload}")

    if rank == 0 and n_gpus == 1 and not args.no_extras:
        also = []
        # single-sequence latency (the literal config: one 12-atom sequence)
        eng = Engine.from_problems([chain_problem(12)], mode="sesolve")
        sec, stats, kms, kl, _ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, 2, 1, None, torch)
        also.append({"workload": "cfg2 single sequence (latency)", "value": T_SEQ_US / sec,
                     "unit": "sim-us/s", "ms_per_sequence": sec * 1e3,
                     "roofline": roofline(12, 1, stats, kms, kl, "k_traj (1 workgroup)")})
        eng.close()
        # quantum-jump trajectories (what Solver.DEFAULT runs for dissipation + stochastic noise)
        mc_prob = chain_problem(12)
        mc_prob["collapse_ops"] = [(float(np.sqrt(2 * 0.05)), "sigma_rr"), (float(np.sqrt(0.02)), "sigma_gr")]
        eng = Engine.from_problems([mc_prob] * 256, mode="mcsolve")
        seeds = np.arange(256, dtype=np.uint64) + np.uint64(12345)
        st = eng.new_state()
        eng.mc_solve(st, [0.0, 0.05], seeds, store=False)
        torch.cuda.synchronize()
        st = eng.new_state()
        tic = time.perf_counter()
        eng.mc_solve(st, [0.0, T_SEQ_US], seeds, store=False)
        torch.cuda.synchronize()
        sec = time.perf_counter() - tic
        also.append({"workload": "256 quantum-jump (mcsolve) trajectories of the 12-atom anneal sequence, "
                                 "dephasing 0.05/us + relaxation 0.02/us",
                     "value": 256 * T_SEQ_US / sec, "unit": "sim-us/s", "ms_per_batch": sec * 1e3,
                     "jumps_per_trajectory": float(eng.mc_jumps().mean()),
                     "kernel": "k_traj<12,1024,1,MC> (persistent, jumps on the device, 1 launch)"})
        eng.close()
        # small noisy registers (the reference's typical mesolve workloads): persistent dm kernel
        ops4 = [(float(np.sqrt(2 * 0.05)), "sigma_rr"), (float(np.sqrt(0.02)), "sigma_gr")]
        eng = Engine.from_problems([chain_problem(4, ops4)] * 256, mode="mesolve")
        sec, stats, kms, kl, _ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, 2, 1, None, torch)
        also.append({"workload": "256 master-equation trajectories of a 4-atom anneal sequence "
                                 "(dephasing + relaxation), one launch",
                     "value": 256 * T_SEQ_US / sec, "unit": "sim-us/s", "ms_per_batch": sec * 1e3,
                     "kernel": "k_traj_dm<4,256,true> (persistent, LDS-resident density matrix)"})
        eng.close()
        # ensemble density matrix of 1024 trajectories (density_matrix_aggregator): fp64 MFMA
        eng = Engine.from_problems([chain_problem(12)] * 1024, mode="sesolve")
        psi = torch.randn(1024, 4096, dtype=torch.complex128, device=eng.device)
        rho = torch.zeros(4096, 4096, dtype=torch.complex128, device=eng.device)
        eng.outer_accumulate(psi, rho)
        torch.cuda.synchronize()
        tic = time.perf_counter()
        for _ in range(5):
            eng.outer_accumulate(psi, rho)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - tic) / 5
        flops = 8.0 * 4096 * 4097 / 2 * 1024  # Hermitian: tiles on or above the diagonal only
        also.append({"workload": "trajectory-averaged density matrix, 1024 x 12-atom kets (rho = sum |psi><psi|)",
                     "value": 1.0 / sec, "unit": "aggregations/s", "ms": sec * 1e3,
                     "roofline": {"bound": "mfma", "kernel": "k_outer_mfma (v_mfma_f64_16x16x4_f64, upper-triangle 64x64 tiles)",
                                  "achieved": flops / sec / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                                  "frac": flops / sec / 1e12 / 78.6, "traffic": None,
                                  "note": "measured issue ceiling of the f64 MFMA on this part: 48 TFLOP/s "
                                          "(tools/ubench/mfma_f64.hip); a full ZGEMM would need 2x the flops"}})
        eng.close()
        # north-star target size, Schroedinger leg: one 14-atom triangular-register sequence
        eng = Engine.from_problems([tri_problem(2, 7)], mode="sesolve")
        t0, t1 = 1.0, 1.1
        sec, stats, kms, kl, _ = timed_run(eng, eng.new_state, t0, t1, 2, 1, None, torch)
        also.append({"workload": "14-atom triangular register, sesolve, single sequence, 100 ns slice at t = 1 us",
                     "value": (t1 - t0) / sec, "unit": "sim-us/s",
                     "passes_per_application": stats["passes"], "taylor_order": stats["last_order"],
                     "roofline": roofline(14, 1, stats, kms, kl, "k_apply<sesolve> (single-launch plan: low bits in LDS, 5 partner tiles from L2; 256 KiB state: launch-latency-bound)")})
        eng.close()
        # ... and a batch of 256 such sequences (state batch = 64 MiB): the streaming regime
        eng = Engine.from_problems([tri_problem(2, 7)] * 256, mode="sesolve")
        t0, t1 = 1.0, 1.02
        sec, stats, kms, kl, _ = timed_run(eng, eng.new_state, t0, t1, 2, 1, None, torch)
        also.append({"workload": "14-atom triangular register, sesolve, 256 sequences, 20 ns slice at t = 1 us",
                     "value": 256 * (t1 - t0) / sec, "unit": "sim-us/s",
                     "passes_per_application": stats["passes"], "taylor_order": stats["last_order"],
                     "roofline": roofline(14, 256, stats, kms, kl, "k_apply14<sesolve> (2^14 register tiles, 1 pass)")})
        eng.close()
        # cfg3: 14-atom triangular register, dephasing Lindblad, HBM-streaming tiled kernel
        ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr")]
        eng = Engine.from_problems([tri_problem(2, 7, ops)], mode="mesolve")
        t0, t1 = 1.0, 1.002
        sec, stats, kms, kl, occ = timed_run(eng, eng.new_state, t0, t1, 2, 1, None, torch)
        also.append({"workload": "cfg3: 14-atom triangular, dephasing mesolve (rho = 4.29 GB), 2 ns slice at t = 1 us",
                     "value": (t1 - t0) / sec, "unit": "sim-us/s", "ms_per_sim_ns": sec * 1e3 / 2,
                     "passes_per_application": stats["passes"], "taylor_order": stats["last_order"],
                     "trace": float(occ[-1].item()),
                     "roofline": roofline(28, 1, stats, kms, kl, "k_apply14<mesolve> + k_symm (Hermitian path: 2^14 register-tile row pass + tile-pair symmetrisation)",
                                          "cfg3:k_apply")})
        eng.close()
        # cfg5: 20-atom sesolve slice
        eng = Engine.from_problems([rect_problem(4, 5)], mode="sesolve")
        t0, t1 = 1.0, 1.02
        sec, stats, kms, kl, occ = timed_run(eng, eng.new_state, t0, t1, 2, 1, None, torch)
        also.append({"workload": "cfg5: 20-atom 4x5 register, sesolve, 20 ns slice at t = 1 us",
                     "value": (t1 - t0) / sec, "unit": "sim-us/s",
                     "passes_per_application": stats["passes"], "taylor_order": stats["last_order"],
                     "roofline": roofline(20, 1, stats, kms, kl, "k_apply<sesolve> (single-launch plan: 2^12 LDS tiles + 8 partner tiles through the Infinity Cache)")})
        eng.close()
        out["also"] = also

    if rank == 0 and n_gpus == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(12)
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
