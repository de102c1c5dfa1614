#!/usr/bin/env python
"""bench.py - sim-us/s of the emulation hot path on MI355X (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload north_star|cfg2|cfg3|cfg4|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = the north-star target (BASELINE.json): the 14-atom triangular register
(2 x 7, spacing R_b), analog Ising anneal of 3100 ns, complex128.

  value            Schroedinger leg.  One "step" = the FULL 3.1 us sequence for a batch of
                   `--batch` (256 = one per CU) independent sequences per GPU; the sequences
                   differ (amplitude / detuning scale factors spread over +-1 %), their tables and
                   initial states are resident in HBM before the timed region.
                   value = GPUs x batch x 3.1 us / seconds-per-step.  Multi-GPU: sequences shard
                   over the ranks, no data-path collective; one all-reduce (RCCL) of the ensemble
                   occupation sums per step.
  single_sequence  ONE 14-atom sequence, full 3.1 us (latency; the one-launch split-operator kernel on one CU).
  lindblad         cfg3: 14-atom dephasing master equation (rho = 4.29 GB), `--lindblad-ns` ns slice
                   (default 100) through the split-operator row path; `--full-lindblad` runs all 3.1 us.
  setup            handle creation + table upload, timed separately (not inside a step).
  roofline         the dominant kernel of the headline leg, timed with HIP events on its launch
                   stream; the bound that binds is named (`valu_f64` for the register-resident
                   kernels, `hbm` for streaming ones) and `frac` <= 1 by construction.
  cpu_baseline     the CPU oracle (SciPy restatement of the QuTiP path) on host cores, rank 0,
                   N = 1 only: primary = one 14-atom sequence on one core; `legs` = the other
                   baselines SURVEY 8(d) lists.
  also             secondary workloads (12-atom batch = the round-1 headline, quantum jumps, small
                   density matrices, ensemble density matrix, cfg4 end to end - 1024 noise trajectories through
                   run_ensemble with and without the density-matrix sum -, cfg5: 20 atoms over the full 3.1 us with
                   the Lanczos figure beside it, 24-atom slice).
  parity_max_abs   sequence 0 of the timed batch against the tight-oracle fixture of the same register (the line is
                   not printed beyond 1e-7).
  collective       N > 1: backend, world size, PCI bus ids of the ranks' devices (all-gathered), an all-reduce of ones.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import sys
import time

# the CPU baselines are single-threaded per process by definition (QuTiP's CSR matvec and zvode are):
# pin the BLAS / OpenMP pools BEFORE numpy loads, or 64 pool workers oversubscribe the host
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
F64_VALU_PEAK = 78.6   # TFLOP/s, fp64 vector peak (256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)
T_SEQ_US = 3.1

# fp64 flops per amplitude per stage of k_ket<14>, counted in the ISA (tools/count_isa.py ->
# profiles/r02_kket_isa.md): per 16 amplitudes of a lane 224 v_fma_f64 + 32 v_mul_f64 + 52..60 v_add_f64
# = 33.25 / 33.75 flops per amplitude per half-stage (second / first half), two half-stages per stage
KKET_FLOPS_PER_AMP_STAGE = 2 * 0.5 * (33.25 + 33.75)
# k_split14_loop<true>: per 32 amplitudes of a lane and stage 1255 v_fma/v_fmac_f64 + 302 v_mul_f64 + 40 v_add_f64
# + 32 v_rndne_f64 in the stage loop of the compiled kernel (tools/count_isa.py split14 -> profiles/r03_ksplit14_isa.md)
KSPLIT14_FLOPS_PER_AMP_STAGE = (2 * 1255 + 302 + 40 + 32) / 32.0
KSPLIT14_NAME = ("k_split14_loop (round-3 register-resident split-operator kernel: two LDS turns per stage; "
                 "set_path(split_turns=True))")
# k_split_reg<14, 5>: per 32 amplitudes of a lane and stage (ONE stage body in the loop) 687 v_fma_f64 + 475 v_fmac_f64
# + 241 v_mul_f64 + 9 v_add_f64 + 7 v_rndne_f64 (tools/count_isa.py splitreg -> profiles/r04_ksplitreg_isa.md)
KSPLITREG_FLOPS_PER_AMP_STAGE = (2 * (687 + 475) + 241 + 9 + 7) / 32.0
# the same stage counted on paper: 14 tan-form rotations (2 FMAs each) + one complex multiplication by the phase
# factor; everything else the kernel spends (building the phase factors, range reduction) is overhead, not work
SPLIT_ALGORITHMIC_FLOPS_PER_AMP_STAGE = 14 * 2 * 2 + 6.0


# ISA counts of the other shapes of the kernel (tools/count_isa.py splitreg N NR; per amplitude and stage).  The plain
# 12-atom kernel runs 16 amplitudes per lane on 256 lanes since round 5 (NR = 4)
KSPLITREG_SHAPES = {12: (4, (2 * 543 + 138 + 8 + 6) / 16.0), 13: (5, (2 * 1098 + 240 + 8 + 7) / 32.0),
                    14: (5, KSPLITREG_FLOPS_PER_AMP_STAGE)}


def split_reg_roofline(n, batch, stats, kms, kl, traffic_key=None, note=None):
    """Roofline of a solve that ran on k_split_reg<n, NR>: fp64 flops by ISA count of that shape (checked for (14, 5) and
    (12, 4) by tests/test_host_logic.py); on paper one atom fewer = one tan-form rotation (2 FMAs) fewer per amplitude."""
    nr, flops = KSPLITREG_SHAPES[n]
    return roofline_valu(2.0**n, batch, stats["n_applications"], flops, kms, kl,
                         KSPLITREG_NAME.replace("<14, 5>", f"<{n}, {nr}>"), traffic_key, note=note,
                         algorithmic_flops_per_amp_stage=SPLIT_ALGORITHMIC_FLOPS_PER_AMP_STAGE - 4.0 * (14 - n))


def ran_split_reg(stats):
    """A solve that took the register-resident split-operator kernel: the controller booked an estimate and a launch
    covered a closed run of stages (the pass-by-pass launches have one launch per stage)."""
    return stats["reserved"][0] > 0.0 and stats["n_launches"] * 20 < stats["n_applications"]
KSPLITREG_NAME = ("k_split_reg<14, 5> (register-resident split-operator kernel: one workgroup per sequence, exact phases x "
                  "single-atom rotations, 6th-order composition over multi-knot sub-steps; lane bits over the DPP crossbar / "
                  "permlane swaps, one chunked LDS pass per stage; one launch per closed run of <= 64 sub-steps)")
# k_traj<12,1024,1>: 120 fp64 instructions per wave and stage for 4 amplitudes per lane
# (profiles/r01_ktraj_counters.md), ~85 % of them FMAs
KTRAJ_FLOPS_PER_AMP_STAGE = 120 * 1.85 / 4.0


KSPLIT_NAME = ("k_split_s (split-operator passes: exact diagonal phase x single-atom rotations in tan form, 2^12-amplitude "
               "register / LDS tiles (2^13 at 21 - 22 atoms), one pass per stage up to 22 atoms; algorithmic bytes = 32 B per "
               "amplitude and stage)")
KAPPLY_NAME = "k_apply<sesolve> (single-launch plan: 2^12 LDS tiles + 8 partner tiles through the Infinity Cache)"


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


def spread_tables(prob, batch: int, spread: float = 0.01, seed: int = 7):
    """Device tables of `batch` DIFFERENT sequences on one register: the shared spline tables of
    `prob` with per-sequence amplitude and detuning scale factors spread over +-spread (what
    amplitude / detuning calibration noise does, hamiltonian_data.py:431-468) - no clones."""
    from pulser_amd.terms import lower

    t = lower([prob])
    desc = np.repeat(t.desc, batch, axis=0)
    rng = np.random.default_rng(seed)
    amp = 1.0 + spread * (2.0 * rng.random(batch) - 1.0)
    det = 1.0 + spread * (2.0 * rng.random(batch) - 1.0)
    amp[0] = det[0] = 1.0  # sequence 0 is the nominal one
    desc["drive_scale"] *= amp[:, None]
    desc["det_scale"] *= det[:, None]
    return dataclasses.replace(t, batch=batch, desc=desc)


def barrier(torch, dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def all_reduce(dist, t, op=None):
    op = op if op is not None else dist.ReduceOp.SUM
    if dist.get_backend() == "gloo":  # RYD_BENCH_BACKEND=gloo: 1-GPU check of the N > 1 path
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)


def timed_run(eng, state_fn, t0, t1, steps, warmup, dist=None, torch=None, **opts):
    """W warm-up + K timed passes over [t0, t1] (no per-launch events inside the timed region),
    then ONE more pass with the library's HIP-event pairs around every launch for the roofline.
    Returns (sec/step, stats of one step, kernel ms of one step, launches of one step, last occ); the state the
    event-timed pass ended in is kept in `timed_run.last_state` (the in-process parity check reads it)."""
    for _ in range(warmup):
        st = state_fn()
        eng.evolve(st, t0, t1, **opts)
        occ = eng.occupations(st).sum(dim=0)  # (the first call of a torch reduction loads its code object: ~10 ms)
        if dist is not None:
            all_reduce(dist, occ)
    barrier(torch, dist)
    eng.reset_stats()
    states = [state_fn() for _ in range(steps)]
    barrier(torch, dist)
    tic = time.perf_counter()
    occ = None
    for st in states:
        eng.evolve(st, t0, t1, **opts)
        occ = eng.occupations(st).sum(dim=0)  # evaluation-time reduction of the state on the device
        if dist is not None:
            all_reduce(dist, occ)  # ensemble sum over ranks (RCCL over xGMI)
    barrier(torch, dist)
    dt = time.perf_counter() - tic
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_reduce(dist, tmax, dist.ReduceOp.MAX)
        dt = float(tmax.item())
    stats = eng.stats()
    for k in ("n_applications", "n_launches", "n_steps"):
        stats[k] //= max(steps, 1)
    timed_run.last_timed_state = states[-1] if states else None  # the state the last TIMED pass produced (parity check)
    del states
    # roofline pass: same work, event-timed
    eng.set_kernel_timing(True)
    st = state_fn()
    eng.evolve(st, t0, t1, **opts)
    torch.cuda.synchronize()
    kms, kl = eng.kernel_timing()
    eng.set_kernel_timing(False)
    timed_run.last_state = st
    return dt / steps, stats, kms, kl, occ


def measured_traffic(key):
    """HBM bytes per launch from the separate rocprofv3 --pmc passes of this round
    (tools/profile.sh -> profiles/*_traffic.json); None if not measured."""
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


def traffic_fields(key):
    """`traffic` = HBM bytes per launch as a plain number (the bench contract), its provenance beside it."""
    t = measured_traffic(key) if key else None
    if t is None:
        return {"traffic": None}
    return {"traffic": t["bytes_per_launch"], "traffic_unit": "bytes per launch", "traffic_source": t["source"],
            "traffic_method": t["method"]}


def roofline_hbm(nb, batch, stats, kernel_ms, launches, kernel_name, traffic_key=None):
    """Streaming kernels: algorithmic bytes = 32 B x 2^nb per generator application."""
    apps = stats["n_applications"]
    bytes_total = 32.0 * (2.0**nb) * batch * apps
    sec = kernel_ms * 1e-3
    achieved = bytes_total / sec if sec > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel_name, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": achieved / HBM_PEAK,
            **traffic_fields(traffic_key),
            "launches": launches, "avg_launch_ms": kernel_ms / max(launches, 1), "applications": apps,
            "us_per_stage": kernel_ms * 1e3 / max(apps, 1),
            "algorithmic_bytes_per_launch": bytes_total / max(launches, 1)}


def roofline_valu(n_amp, rows, stages, flops_per_amp_stage, kernel_ms, launches, kernel_name,
                  traffic_key=None, note=None, algorithmic_flops_per_amp_stage=None):
    """Register / LDS-resident kernels: the state never streams through HBM, the fp64 vector pipe
    binds.  achieved = fp64 flops the kernel ISSUES (ISA count per amplitude per stage) / kernel time; the bare
    algorithmic count (what the stage needs on paper) and the time per stage stand beside it, because a cheaper
    evaluation lowers `frac` while making the kernel faster."""
    flops = flops_per_amp_stage * n_amp * rows * stages
    sec = kernel_ms * 1e-3
    tf = flops / sec / 1e12 if sec > 0 else 0.0
    out = {"bound": "valu_f64", "kernel": kernel_name, "achieved": tf, "peak": F64_VALU_PEAK,
           "unit": "TFLOP/s", "frac": tf / F64_VALU_PEAK,
           **traffic_fields(traffic_key),
           "launches": launches, "avg_launch_ms": kernel_ms / max(launches, 1),
           "stages": stages, "us_per_stage": kernel_ms * 1e3 / max(stages, 1),
           "flops_per_amplitude_per_stage": flops_per_amp_stage,
           "isa_flops_per_launch": flops / max(launches, 1),
           "hbm_equivalent_GBps": 32.0 * n_amp * rows * stages / sec / 1e9 if sec > 0 else 0.0,
           # what HBM has to carry for a launch on paper: every ket once in and once out (16 B per amplitude each way)
           "algorithmic_bytes_per_launch": 32.0 * n_amp * rows}
    if algorithmic_flops_per_amp_stage:
        alg = algorithmic_flops_per_amp_stage * n_amp * rows * stages
        out["algorithmic_flops_per_amplitude_per_stage"] = algorithmic_flops_per_amp_stage
        out["algorithmic_flops_per_launch"] = alg / max(launches, 1)
        out["frac_algorithmic"] = alg / sec / 1e12 / F64_VALU_PEAK if sec > 0 else 0.0
    if note:
        out["note"] = note
    return out


# ----------------------------------------------------------------------------- the driver's line
DETAIL_PREFIX = "BENCH_DETAIL "
MAX_LINE_BYTES = 6144   # the driver keeps an 8-KB tail of stdout: the contract line must fit with room to spare
_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_algorithmic", "us_per_stage", "traffic",
              "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "isa_flops_per_launch", "launches",
              "avg_launch_ms", "hbm_equivalent_GBps")
_CFG_KEYS = ("ms_min", "ms_median", "ms_max", "spread", "n_atoms", "sequences_per_gpu", "sim_us_per_sequence", "stages_per_sequence", "parity_max_abs",
             "single_sequence_sim_us_per_s", "lindblad_seconds", "lindblad_sim_us_per_s", "lindblad_roofline_frac",
             "n_trajectories", "sim_us_per_s", "n_measures", "histogram_total", "passes_per_application", "order",
             "generator_applications_per_sequence")


def _short(text, limit):
    text = str(text)
    return text if len(text) <= limit else text[: limit - 3] + "..."


def _num(v):
    """Numbers of the line at 6 significant digits (the detail file keeps every digit)."""
    if isinstance(v, float) and v == v and abs(v) != float("inf") and not (abs(v) < 1e15 and v == int(v)):
        return float(f"{v:.6g}")
    return v


def _find_leg(also, start):
    for leg in also or []:
        if str(leg.get("workload", "")).startswith(start):
            return leg
    return None


def driver_line(out: dict) -> dict:
    """The ONE short JSON line of the driver contract, cut from the full result `out` (which goes to
    bench_detail.json): scalar fields only, names instead of descriptions, no notes.  Pure function of `out` (a CPU
    test feeds it a canned dict); <= MAX_LINE_BYTES by construction of the key lists, asserted where it is printed."""
    line = {k: _num(out[k]) for k in _TOP_KEYS if k in out}
    cfg_in = out.get("config") or {}
    cfg = {"workload": _short(cfg_in.get("workload", ""), 200)}
    cfg.update({k: _num(cfg_in[k]) for k in _CFG_KEYS if k in cfg_in})
    if "parallelism" in cfg_in:
        cfg["parallelism"] = _short(cfg_in["parallelism"], 60).split(" (")[0]
    api = cfg_in.get("api_end_to_end") or {}
    for spec, key in (("Full", "api_full_ms"), ("Minimal", "api_minimal_ms")):
        if spec in api:
            cfg[key] = _num(api[spec]["ms"])
    also = out.get("also")
    for start, key, field in (("cfg2:", "cfg2_sim_us_per_s", "value"), ("cfg4:", "cfg4_traj_per_s", "value"),
                              ("cfg5: 20-atom", "cfg5_sim_us_per_s", "value"),
                              ("f-1:", "multilevel_sim_us_per_s", "value"), ("f-4:", "xy_sim_us_per_s", "value")):
        leg = _find_leg(also, start)
        if leg is not None:
            cfg[key] = _num(leg[field])
    line["config"] = cfg
    roof = out.get("roofline")
    if roof:
        r = {k: _num(roof[k]) for k in _ROOF_KEYS if k in roof}
        r["kernel"] = _short(str(roof.get("kernel", "")).split(" (")[0], 80)  # the name only
        r.setdefault("traffic", None)
        line["roofline"] = r
    else:
        line["roofline"] = None
    cpu = out.get("cpu_baseline")
    if cpu:
        c = {k: _num(cpu[k]) for k in ("value", "unit", "cores", "kind", "host_cpu_count") if k in cpu}
        c["sample"] = _short(cpu.get("sample", ""), 120)
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    col = out.get("collective")
    if col:
        line["collective"] = {k: col[k] for k in ("backend", "world_size", "distinct_devices", "allreduce_of_ones")
                              if k in col}
    line["detail"] = "bench_detail.json"
    return line


# ----------------------------------------------------------------------------- CPU baselines
def _oracle_sesolve_time(prob, t_end, reps=1):
    from oracle import qutip_path as qp

    ham = qp.build_hamiltonian(prob)
    n = prob["n_qudits"]
    psi0 = qp.all_ground_state(n, prob["eigenbasis"])
    s = prob["samples"]["Global"]["ground-rydberg"]
    opts = qp.default_options([(s["amp"], s["det"])], 3100)
    counter = [0]
    tic = time.perf_counter()
    for _ in range(reps):
        qp.sesolve(ham, psi0, np.array([0.0, t_end]), counter=counter, **opts)
    return (time.perf_counter() - tic) / reps, counter[0] // reps


def _oracle_mesolve_time(prob, t_end):
    from oracle import qutip_path as qp

    ham = qp.build_hamiltonian(prob)
    psi0 = qp.all_ground_state(prob["n_qudits"], prob["eigenbasis"])
    s = prob["samples"]["Global"]["ground-rydberg"]
    opts = qp.default_options([(s["amp"], s["det"])], 3100)
    counter = [0]
    tic = time.perf_counter()
    qp.mesolve(ham, psi0, np.array([0.0, t_end]), counter=counter, **opts)
    return time.perf_counter() - tic, counter[0]


def _pool_warm(_):
    os.environ["OMP_NUM_THREADS"] = "1"
    _oracle_sesolve_time(chain_problem(12), 0.01)
    return 0


def _pool_trajectory(seed):
    """One noisy 12-atom trajectory on one core (cfg4 CPU baseline worker)."""
    os.environ["OMP_NUM_THREADS"] = "1"
    rng = np.random.default_rng(seed)
    prob = chain_problem(12)
    s = prob["samples"]["Global"]["ground-rydberg"]
    s["amp"] = s["amp"] * max(0.0, rng.normal(1.0, 0.05))
    s["det"] = s["det"] + rng.normal(0.0, 0.3)
    dt, _ = _oracle_sesolve_time(prob, T_SEQ_US)
    return dt


def cpu_baselines(full: bool):
    """Oracle (kind 'port') timed on the box's host cores.  Bounded: ~10 s primary + ~60 s of legs."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    ncpu = os.cpu_count() or 1
    dt14, rhs14 = _oracle_sesolve_time(tri_problem(2, 7), T_SEQ_US)
    out = {"value": T_SEQ_US / dt14, "unit": "sim-us/s", "cores": 1, "kind": "port",
           "sample": f"one 14-atom sequence (3.1 us), zvode Adams at QuTiP defaults on SciPy CSR terms: {rhs14} RHS, {dt14:.1f} s",
           "sample_detail": "the north-star sequence on the triangular register; SciPy CSR terms + not-a-knot spline + "
                            "zvode Adams (atol 1e-8, rtol 1e-6, max_step 1 ns), one core",
           "host_cpu_count": ncpu}
    if not full:
        return out
    # the drop-in call's evaluation-time lists (config.api_end_to_end): the same oracle returning a state at every
    # sample ("Full", the reference's default, simulation.py:137), at every 10th (0.1) - "Minimal" is the primary above
    from oracle import qutip_path as qp

    prob = tri_problem(2, 7)
    ham = qp.build_hamiltonian(prob)
    psi0 = qp.all_ground_state(14, prob["eigenbasis"])
    sg = prob["samples"]["Global"]["ground-rydberg"]
    opts = qp.default_options([(sg["amp"], sg["det"])], 3100)
    grid = np.arange(3101) * 1e-3
    api = {"Minimal": dt14}
    for name, tl in (("Full", grid), ("0.1", grid[np.linspace(0, 3100, 310, dtype=int)])):
        tl = np.union1d(tl, [0.0, T_SEQ_US])
        tic = time.perf_counter()
        qp.sesolve(ham, psi0, tl, **opts)
        api[name] = time.perf_counter() - tic
    out["api_eval_times_seconds"] = api
    legs = []
    # sesolve scaling with N (full 3.1 us at 12, slice at 16), one core
    dt12, rhs12 = _oracle_sesolve_time(chain_problem(12), T_SEQ_US)
    dt16, rhs16 = _oracle_sesolve_time(rect_problem(4, 4), 0.2)
    dt16 *= T_SEQ_US / 0.2
    k = np.polyfit([12, 14, 16], np.log2([dt12, dt14, dt16]), 1)
    legs.append({"workload": "sesolve, one core, full 3.1 us (16 atoms: 200 ns slice scaled)", "unit": "sim-us/s",
                 "n_atoms": [12, 14, 16], "value": [T_SEQ_US / dt12, T_SEQ_US / dt14, T_SEQ_US / dt16],
                 "fitted_time_doubling_per_atom": float(k[0]),
                 "extrapolated_20_atoms_sim_us_per_s": float(T_SEQ_US / 2 ** np.polyval(k, 20)), "cores": 1})
    # mesolve (cfg3 physics) at reduced N: the explicit-operator path does not fit at 14 atoms
    ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr")]
    dm = []
    for rows_cols, t_end in (((2, 3), 0.2), ((2, 4), 0.05)):
        dt, rhs = _oracle_mesolve_time(tri_problem(*rows_cols, ops), t_end)
        dm.append((2 * rows_cols[1], t_end / dt))
    k2 = np.polyfit([d[0] for d in dm], np.log2([1.0 / d[1] for d in dm]), 1)
    legs.append({"workload": "mesolve dephasing (cfg3 physics), one core, slices of 200 / 50 ns",
                 "unit": "sim-us/s", "n_atoms": [d[0] for d in dm], "value": [d[1] for d in dm],
                 "fitted_time_doubling_per_atom": float(k2[0]),
                 "extrapolated_14_atoms_sim_us_per_s": float(1.0 / 2 ** np.polyval(k2, 14)), "cores": 1})
    # cfg4: all host cores, one trajectory per process
    from multiprocessing import get_context

    nproc = min(ncpu, 64)
    n_traj = 2 * nproc
    with get_context("fork").Pool(nproc) as pool:
        pool.map(_pool_warm, range(nproc))  # imports and first-call overheads outside the timed map
        tic = time.perf_counter()
        pool.map(_pool_trajectory, range(n_traj), chunksize=1)
        wall = time.perf_counter() - tic
    legs.append({"workload": "cfg4: 12-atom noisy trajectories, process pool, one trajectory per process",
                 "unit": "trajectories/s", "value": n_traj / wall, "cores": nproc, "n_trajectories": n_traj,
                 "sim_us_per_s": n_traj * T_SEQ_US / wall, "wall_s": wall})
    out["legs"] = legs
    return out


def cfg4_line(n_traj, steps, warmup, dist, torch, n_gpus, common):
    """BASELINE configs[3]: noise trajectories of the 12-atom sequence, END TO END through the emulator front-end (noise
    draws on rank 0, factored lowering, solve, reference-order sampling), sharded over the ranks with one all-reduce of
    the histograms (strong scaling); the same again with the ensemble density matrix (A15 on the device)."""
    from pulser_amd import NoiseModel, QutipEmulator, problem as P
    from pulser_amd.distributed import run_ensemble
    from pulser_amd.hamiltonian_data import single_global_channel

    coords = P.register_coords(P.square_rect(1, 12), blockade_radius())
    smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
    inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
    nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.005,
                    p_false_pos=0.01, p_false_neg=0.05)

    def one_pass(seed, density_matrix=False):
        np.random.seed(seed)
        emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=n_traj, evaluation_times="Minimal")
        # blocks of 512 trajectories: two 12-atom kets per CU (the register-resident kernel runs 41 300 sim-us/s at 512
        # sequences against 29 700 at 256, profiles/r05_size_batch_map.md)
        return run_ensemble(emu, dist=dist, batch=int(os.environ.get("RYD_CFG4_BATCH", "512")), density_matrix=density_matrix)

    for w in range(warmup):
        one_pass(100 + w)
    barrier(torch, dist)
    tic = time.perf_counter()
    each = []
    for k in range(steps):
        t_k = time.perf_counter()
        res = one_pass(k)
        torch.cuda.synchronize()
        each.append((time.perf_counter() - t_k) * 1e3)
    barrier(torch, dist)
    sec = (time.perf_counter() - tic) / steps
    if dist is not None:
        tmax = torch.tensor([sec], dtype=torch.float64, device="cuda")
        all_reduce(dist, tmax, dist.ReduceOp.MAX)
        sec = float(tmax.item())
    # the same with the ensemble density matrix (density_matrix_aggregator: the 268-MB mean of |psi><psi|
    # at both evaluation times, formed on the device and all-reduced there)
    one_pass(200, True)
    barrier(torch, dist)
    tic = time.perf_counter()
    res_dm = one_pass(0, True)
    barrier(torch, dist)
    sec_dm = time.perf_counter() - tic
    tr_dm = float(torch.diagonal(res_dm["density_matrices"][-1]).real.sum().item())
    del res_dm
    return {"metric": "noise trajectories/s, 12-atom anneal sequence, end to end (draws, lowering, sesolve, sampling)",
            "value": n_traj / sec, **common, "unit": "trajectories/s", "scaling": "strong",
            "ms_per_step": sec * 1e3,
            "config": {"workload": "BASELINE configs[3]: 12-atom register, 1024 noise trajectories "
                                   "(doppler + amplitude + SPAM), sharded over the ranks, one all-reduce "
                                   "of the bitstring histograms", "n_atoms": 12, "n_trajectories": n_traj,
                       "sim_us_per_s": n_traj * T_SEQ_US / sec, "n_measures": int(res["n_measures"]),
                       "histogram_total": int(res["histograms"].sum()),
                       # run-to-run spread of the end-to-end ensemble on this rank
                       "ms_min": float(np.min(each)), "ms_median": float(np.median(each)), "ms_max": float(np.max(each)),
                       "spread": float((np.max(each) - np.min(each)) / np.median(each)),
                       "mean_occupations_final": [float(v) for v in res["mean_occupations"][-1]],
                       "with_density_matrix": {"ms_per_step": sec_dm * 1e3, "ratio": sec_dm / sec,
                                               "trace_final": tr_dm,
                                               "note": "A15 on the device: ryd_outer_accumulate_dim per batch and "
                                                       "evaluation time, all-reduce of the device tensor"},
                       "parallelism": f"dp{n_gpus} over trajectories"},
            "roofline": None}


def api_end_to_end(torch, cpu_seconds=None):
    """What a drop-in user gets: wall clock of ``QutipEmulator(<the north-star sequence>).run()``
    - construction, lowering, handle + upload, solve, snapshots, result objects - for the reference's default
    ``evaluation_times="Full"`` (simulation.py:137, 961), "Minimal" and 0.1.  Second call of each (the first call of a
    process also loads code objects: `first_call_ms`); the final state of every call against the tight-oracle fixture."""
    import warnings

    from pulser_amd import QutipEmulator, problem as P
    from pulser_amd.hamiltonian_data import single_global_channel

    coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
    smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
    inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
    _, fx = P.load_problem(os.path.join(ROOT, "tests", "golden", "ns_tri14_anneal.npz"))
    ref_final = np.asarray(fx["oracle_states_tight"])[-1]
    out = {"workload": "pulser_amd.QutipEmulator(<14-atom triangular register, anneal 3100 ns>, evaluation_times=...).run(): "
                       "wall clock of the whole call (construction, lowering, handle + upload, solve, snapshots, "
                       "CoherentResults) + reading the final state; one sequence",
           "unit": "sim-us/s"}
    for spec in ("Full", "Minimal", 0.1):
        rec = {}
        for rep in range(3):
            torch.cuda.synchronize()
            tic = time.perf_counter()
            emu = QutipEmulator(inputs, evaluation_times=spec)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", DeprecationWarning)
                res = emu.run()
            final = np.asarray(res.states[-1])[:, 0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - tic
            if rep == 0:
                rec["first_call_ms"] = dt * 1e3
            else:
                rec["ms"] = min(rec.get("ms", 1e30), dt * 1e3)
        st = emu.last_engine_stats
        rec.update({"value": T_SEQ_US / (rec["ms"] * 1e-3), "n_evaluation_times": int(len(emu.evaluation_times)),
                    "stages": st["n_applications"], "launches": st["n_launches"], "local_error_estimate": st["reserved"][0],
                    "final_state_max_abs_vs_tight_oracle": float(np.max(np.abs(final - ref_final)))})
        if cpu_seconds and str(spec) in cpu_seconds:
            rec["cpu_oracle_s"] = cpu_seconds[str(spec)]
            rec["cpu_oracle_sim_us_per_s"] = T_SEQ_US / cpu_seconds[str(spec)]
        # reading EVERY stored state back (what plotting an observable over the evaluation times costs on top)
        import gc

        gc.collect()  # (the previous call's 813 MB of snapshots are released outside the timed loop)
        tic = time.perf_counter()
        acc = 0.0
        for stt in res.states:
            acc += float(abs(np.asarray(stt)[0, 0]))
        rec["read_all_states_ms"] = (time.perf_counter() - tic) * 1e3
        out[str(spec)] = rec
        del res, emu
    out["full_over_minimal"] = out["Full"]["ms"] / out["Minimal"]["ms"]
    return out


def device_copy_bandwidth(torch):
    """Measured device-to-device copy bandwidth (read + write bytes / time), 2 GiB buffer."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    a.fill_(1.0)
    b.copy_(a)
    torch.cuda.synchronize()
    tic = time.perf_counter()
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - tic) / 5
    del a, b
    return 2.0 * n * 8 / sec / 1e9


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="independent sequences per GPU")
    ap.add_argument("--workload", default="north_star")
    ap.add_argument("--slice-ns", type=int, default=2, help="cfg3/cfg5 workloads: simulated ns per step")
    ap.add_argument("--atoms", type=int, default=20, choices=[20, 22, 24], help="cfg5 workload: register size")
    ap.add_argument("--method", default="auto", choices=["auto", "taylor", "krylov", "split"],
                    help="cfg5 workload: propagator (auto = split-operator passes)")
    ap.add_argument("--lindblad-ns", type=int, default=0,
                    help="north_star: cfg3 leg over a slice of this many ns at t = 1 us (0 = the FULL 3.1 us, the default)")
    ap.add_argument("--full-lindblad", action="store_true", help="(kept for old recipes: the full leg is the default)")
    ap.add_argument("--trajectories", type=int, default=1024, help="cfg4 workload: noise trajectories")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="north_star: headline leg only (profiling runs)")
    ap.add_argument("--no-ket", action="store_true", help="disable k_ket / the split-operator rows (A/B runs)")
    ap.add_argument("--no-split14", action="store_true",
                    help="north star: keep the batch on k_ket instead of the split-operator kernel (A/B runs)")
    ap.add_argument("--split-turns", action="store_true",
                    help="north star: the round-3 kernel k_split14_loop instead of k_split_reg (A/B runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_result = None
    if rank == 0 and world_env == 1 and not args.no_cpu and args.workload == "north_star":
        # host-core baselines first: the process pool forks before any HIP context exists
        cpu_result = cpu_baselines(full=not args.no_extras)

    import torch

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # N > 1 always; a one-rank job under torch.distributed.run with RYD_BENCH_BACKEND set runs the same collective code
    # at world size 1 (legal for RCCL): how the nccl branch is executed on a one-GPU box (tests/test_gpu_bench.py)
    if world > 1 or ("RANK" in os.environ and os.environ.get("RYD_BENCH_BACKEND")):
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
    collective = None
    if dist is not None:
        # what the ranks really ran on: backend, world size, the PCI bus id of every rank's device (all-gathered), and
        # one all-reduce of ones through the backend (its result must be the world size)
        prop = torch.cuda.get_device_properties(torch.cuda.current_device())
        ident = {"rank": rank, "device": torch.cuda.current_device(), "name": prop.name,
                 "pci": "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0),
                                            getattr(prop, "pci_device_id", 0)),
                 "uuid": str(getattr(prop, "uuid", ""))}
        gathered = [None] * world
        dist.all_gather_object(gathered, ident)
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        all_reduce(dist, ones)
        collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": gathered,
                      "distinct_devices": len({(g["pci"], g["uuid"]) for g in gathered}),
                      "allreduce_of_ones": float(ones.item())}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")

    from pulser_amd.engine import Engine

    n_gpus = max(world, 1)
    out = {}
    common = {"n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "f64", "data": "synthetic", "unit": "sim-us/s"}
    extras_ok = rank == 0 and n_gpus == 1 and not args.no_extras

    if args.workload == "north_star":
        n, B = 14, args.batch
        tic = time.perf_counter()
        tables = spread_tables(tri_problem(2, 7), B)
        lower_s = time.perf_counter() - tic
        tic = time.perf_counter()
        eng = Engine(tables, mode="sesolve")
        torch.cuda.synchronize()
        create_s = time.perf_counter() - tic
        # ... and once more (what every later handle of the process costs: no code-object load, a warm allocator)
        tic = time.perf_counter()
        eng_warm = Engine(tables, mode="sesolve")
        torch.cuda.synchronize()
        create_warm_s = time.perf_counter() - tic
        eng_warm.close()
        if args.no_ket or args.no_split14 or args.split_turns:
            eng.set_path(False, no_ket=args.no_ket, no_split14=args.no_split14, split_turns=args.split_turns)
        sec, stats, kms, kl, occ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, args.steps, args.warmup, dist, torch)
        value = n_gpus * B * T_SEQ_US / sec
        ket = stats["n_launches"] == 1
        split14 = not ket and not args.no_ket and stats["reserved"][0] > 0.0
        # in-process parity: sequence 0 of the batch is the nominal sequence = the tight-oracle fixture of the headline
        # register (tests/golden/ns_tri14_anneal.npz: zvode rtol 1e-13, final time); the line is not printed when the
        # state the timed kernels produced is further than the stated bar (1e-7) from it
        from pulser_amd import problem as P_

        _, fx = P_.load_problem(os.path.join(ROOT, "tests", "golden", "ns_tri14_anneal.npz"))
        ref_final = np.asarray(fx["oracle_states_tight"])[-1]
        parity = float(np.max(np.abs(timed_run.last_timed_state[0].cpu().numpy() - ref_final)))
        if not parity < 1e-7:
            raise SystemExit(f"bench.py: sequence 0 of the timed batch is {parity:.3e} from the tight oracle (bar 1e-7)")
        # every rank runs the same batch of sequences, so this number must not depend on the number of GPUs
        ens = [float(v) / (n_gpus * B) for v in occ.cpu().numpy()]
        out = {
            "metric": "sim-us/sec, 14-atom Rydberg anneal sequence, sesolve fp64 (aggregate over independent sequences)",
            "value": value, **common, "ms_per_step": sec * 1e3,
            "config": {
                "workload": "north star: 14-atom triangular register (2 x 7 at R_b), analog Ising anneal 3100 ns, sesolve "
                            "complex128; step = the full 3.1 us for a batch of different sequences per GPU",
                "workload_detail": "BASELINE.json north_star; the sequences of a batch differ (amplitude / detuning scale "
                                   "factors spread +-1 %), tables and initial states resident in HBM before the timed region",
                "n_atoms": n, "sequences_per_gpu": B, "sim_us_per_sequence": T_SEQ_US,
                "integrator": "CF4 Magnus; exponentials by the in-place symplectic scheme (k_ket), "
                              "2e-11 per exponential" if ket else
                              "split-operator: exact diagonal phases x exact single-atom rotations, 6th-order 10-stage "
                              "composition over sub-steps of <= 8 knot intervals, measured step-size control "
                              "(accumulated local-error estimate %.1e; %s)" % (stats["reserved"][0], "k_split14_loop" if args.split_turns else "k_split_reg") if split14
                              else "CF4 Magnus + Taylor(Horner), 1e-10 per exponential",
                "stages_per_sequence": stats["n_applications"], "cf4_steps": stats["n_steps"],
                "parallelism": f"dp{n_gpus} (independent sequences shard over ranks; all-reduce of ensemble sums only)",
            },
            "ensemble_mean_occupations": ens[:-1], "ensemble_mean_norm": ens[-1],
            "parity_max_abs": parity,
            "parity_reference": "tests/golden/ns_tri14_anneal.npz (tight oracle: zvode rtol 1e-13), final state, sequence 0 "
                                "of the batch the LAST TIMED step produced; bar 1e-7",
            "setup": {"lowering_ms": lower_s * 1e3, "handle_and_upload_ms": create_s * 1e3,
                      "handle_and_upload_warm_ms": create_warm_s * 1e3,
                      "warm_setup_over_step": create_warm_s / sec,
                      "note": "spline lowering on the host + ryd_create / ryd_set_* uploads (coefficient tables, "
                              "descriptors, interaction matrix -> E0 on the device); `handle_and_upload_ms` is the FIRST "
                              "handle of the process (code-object load, first allocations), `..._warm_ms` the second on the "
                              "same tables = what a step would pay with the upload inside (SURVEY 8d); the step includes the "
                              "evaluation-time occupation reduction and its all-reduce.  config.api_end_to_end times whole "
                              "front-end calls, every upload inside"},
        }
        # driver-visible copies (the driver keeps metric / value / config / roofline / cpu_baseline)
        out["config"]["parity_max_abs"] = parity
        out["config"]["setup_warm_ms"] = create_warm_s * 1e3
        out["config"]["value_with_warm_setup_inside_step"] = n_gpus * B * T_SEQ_US / (sec + create_warm_s)
        if ket:
            out["roofline"] = roofline_valu(
                2.0**n, B, stats["n_applications"], KKET_FLOPS_PER_AMP_STAGE, kms, kl,
                "k_ket<14> (register-resident ket, in-place symplectic exponential; one launch per step)",
                "north_star:k_ket",
                note="the state lives in registers / LDS for the whole sequence; HBM sees the initial load, "
                     "the final store and the tables only ('traffic'). hbm_equivalent_GBps = what a "
                     "streaming kernel would have to sustain for the same applications")
        elif split14:
            out["roofline"] = roofline_valu(
                2.0**n, B, stats["n_applications"],
                KSPLIT14_FLOPS_PER_AMP_STAGE if args.split_turns else KSPLITREG_FLOPS_PER_AMP_STAGE, kms, kl,
                KSPLIT14_NAME if args.split_turns else KSPLITREG_NAME,
                "north_star:k_split14" if args.split_turns else "north_star:k_split_reg",
                note="the kets live in registers for a closed run of sub-steps; HBM sees the states once per launch in "
                     "and out plus the per-stage coefficients ('traffic'). Stages include the step-size controller's "
                     "check sub-steps. `frac` counts the fp64 instructions the kernel issues (ISA), `frac_algorithmic` "
                     "the rotations + one complex multiplication per amplitude a stage needs on paper. Earlier kernels "
                     "of this line: k_split14_loop (round 3: 12.2 us per stage, set_path(split_turns=True)), k_ket "
                     "(26 253 stages, set_path(no_split14=True))",
                algorithmic_flops_per_amp_stage=SPLIT_ALGORITHMIC_FLOPS_PER_AMP_STAGE)
        else:
            out["roofline"] = roofline_hbm(n, B, stats, kms, kl, "k_apply14<sesolve> (2^14 register tiles, 1 pass)")
        eng.close()

        if rank == 0 and n_gpus == 1 and not args.no_legs:
            # latency: ONE sequence, full length
            eng = Engine.from_problems([tri_problem(2, 7)], mode="sesolve")
            if args.no_ket:
                eng.set_path(False, no_ket=True)
            s1, st1, k1, l1, _ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, 1, 1, None, torch)
            out["single_sequence"] = {
                "workload": "one 14-atom triangular-register sequence, full 3.1 us, sesolve", "value": T_SEQ_US / s1,
                "unit": "sim-us/s", "ms_per_sequence": s1 * 1e3, "stages": st1["n_applications"],
                "launches": st1["n_launches"],
                "roofline": roofline_valu(2.0**n, 1, st1["n_applications"], KKET_FLOPS_PER_AMP_STAGE, k1, l1,
                                          "k_ket<14> (one workgroup = one CU of 256)",
                                          note="a single sequence occupies one CU; frac is against the whole chip")
                if st1["n_launches"] == 1 else
                roofline_valu(2.0**n, 1, st1["n_applications"], KSPLITREG_FLOPS_PER_AMP_STAGE, k1, l1,
                              KSPLITREG_NAME + "; one workgroup = one CU of 256",
                              note="a single sequence occupies one CU; frac is against the whole chip (x 256 = the "
                                   "fraction of that CU's fp64 pipe)",
                              algorithmic_flops_per_amp_stage=SPLIT_ALGORITHMIC_FLOPS_PER_AMP_STAGE)
                if st1["n_launches"] * 20 < st1["n_applications"] else
                roofline_hbm(n, 1, st1, k1, l1, KSPLIT_NAME + "; 4 tiles: launch-latency-bound"),
                "local_error_estimate": st1["reserved"][0]}
            eng.close()
            # Lindblad leg (cfg3)
            ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr")]
            eng = Engine.from_problems([tri_problem(2, 7, ops)], mode="mesolve")
            if args.no_ket or os.environ.get("RYD_BENCH_ROWS_KET"):
                eng.set_path(False, no_ket=args.no_ket, rows_ket=bool(os.environ.get("RYD_BENCH_ROWS_KET")))
            args.full_lindblad = args.full_lindblad or args.lindblad_ns <= 0
            if args.full_lindblad:
                t0, t1 = 0.0, T_SEQ_US
            else:
                t0, t1 = 1.0, 1.0 + 1e-3 * args.lindblad_ns
            # one block of warm-up on a scratch state (work buffers, code objects), then the timed full-length run
            warm = eng.new_state()
            eng.evolve(warm, 0.0, 0.004)
            del warm
            sl, stl, kl_ms, kl_n, occl = timed_run(eng, eng.new_state, t0, t1, 1, 0, None, torch)
            leg = {"workload": f"cfg3: 14-atom triangular register, dephasing 0.05/us master equation "
                               f"(rho = 4.29 GB), {'full 3.1 us' if args.full_lindblad else f'{args.lindblad_ns} ns slice at t = 1 us'}",
                   "value": (t1 - t0) / sl, "unit": "sim-us/s", "ms_per_sim_ns": sl * 1e3 / ((t1 - t0) * 1e3),
                   "seconds": sl, "trace": float(occl[-1].item()), "launches": stl["n_launches"],
                   "cf4_steps": stl["n_steps"], "stages": stl["n_applications"],
                   "extrapolated_full_sequence_s": sl * T_SEQ_US / (t1 - t0)}
            if not args.no_ket:
                # two row passes per conjugation, each a full ket stage on 2^14 rows of 2^14 amplitudes
                rows_ket = bool(os.environ.get("RYD_BENCH_ROWS_KET"))
                leg["integrator"] = ("4th-order operator splitting (Chin 4A + exact commutator kick), blocks of "
                                     + ("2 + 2" if rows_ket else "4 + 4") + " knot intervals; "
                                     "U rho U^+ as two row passes + one conjugate transposition; the unitary of a half "
                                     "block by " + ("CF4 steps on k_ket (round 3)" if rows_ket else
                                                    "split-operator sub-steps (6th-order 10-stage composition over the four knots where the "
                                                    "waveforms are one polynomial, 4th-order 6-stage ones elsewhere) on k_split_reg"))
                leg["roofline"] = roofline_valu(
                    2.0**n, 2.0**n * 2, stl["n_applications"], KKET_FLOPS_PER_AMP_STAGE if rows_ket else KSPLITREG_FLOPS_PER_AMP_STAGE,
                    kl_ms, kl_n, ("k_ket<14> row passes" if rows_ket else "k_split_reg<14, 5, ROWS> row passes (persistent "
                                  "workgroups, 64 rows each)") + " (+ k_transpose_conj)", "cfg3:k_ket" if rows_ket else "cfg3:k_split_reg",
                    note="kernel time includes the transpositions (HBM-bound, 8.6 GB each) and the kick stages",
                    algorithmic_flops_per_amp_stage=None if rows_ket else SPLIT_ALGORITHMIC_FLOPS_PER_AMP_STAGE)
            else:
                leg["roofline"] = roofline_hbm(28, 1, stl, kl_ms, kl_n,
                                               "k_apply14<mesolve> + k_symm (Hermitian path)", "cfg3:k_apply")
            out["lindblad"] = leg
            eng.close()
            out["config"]["single_sequence_sim_us_per_s"] = out["single_sequence"]["value"]
            out["config"]["lindblad_seconds"] = leg["seconds"]
            out["config"]["lindblad_sim_us_per_s"] = leg["value"]
            out["config"]["lindblad_trace"] = leg["trace"]
            out["config"]["lindblad_roofline_frac"] = leg["roofline"]["frac"]
            out["config"]["api_end_to_end"] = api_end_to_end(
                torch, (cpu_result or {}).get("api_eval_times_seconds"))

    elif args.workload == "cfg2":
        n, B = 12, args.batch
        eng = Engine.from_problems([chain_problem(n)] * B, mode="sesolve")
        sec, stats, kms, kl, occ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, args.steps, args.warmup, dist, torch)
        out = {
            "metric": "sim-us/sec, 12-atom Rydberg anneal sequence, sesolve fp64 (aggregate over sequences)",
            "value": n_gpus * B * T_SEQ_US / sec, **common, "ms_per_step": sec * 1e3,
            "config": {"workload": "BASELINE configs[1]: 12-atom chain at the blockade radius, analog Ising "
                                   "anneal 3100 ns, sesolve complex128; batch of independent sequences per GPU",
                       "n_atoms": n, "sequences_per_gpu": B, "sim_us_per_sequence": T_SEQ_US,
                       "integrator": "split-operator, 6th-order composition over multi-knot sub-steps (k_split_reg<12, 4>)"
                                     if ran_split_reg(stats) else "CF4 Magnus + Taylor(Horner), tol 1e-10/exponential",
                       "generator_applications_per_sequence": stats["n_applications"],
                       "parallelism": f"dp{n_gpus} (independent sequences, all-reduce of ensemble sums only)"},
            "roofline": split_reg_roofline(n, B, stats, kms, kl, "cfg2:k_split_reg") if ran_split_reg(stats) else
                        roofline_valu(2.0**n, B, stats["n_applications"], KTRAJ_FLOPS_PER_AMP_STAGE, kms, kl,
                                      "k_traj<12,1024,1> (persistent, LDS-resident)", "cfg2:k_traj"),
        }
        eng.close()
    elif args.workload == "cfg4":
        out = cfg4_line(args.trajectories, args.steps, args.warmup, dist, torch, n_gpus, common)
        extras_ok = False
        args.no_cpu = True
    elif args.workload in ("cfg3", "cfg5"):
        # streaming workloads as the primary line (used for the rocprofv3 passes)
        if args.workload == "cfg3":
            ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr")]
            eng = Engine.from_problems([tri_problem(2, 7, ops)], mode="mesolve")
            if args.no_ket:
                eng.set_path(False, no_ket=True)
            t0, t1, nb = 1.0, 1.0 + 1e-3 * args.slice_ns, 28
            kname = ("k_apply14<mesolve> + k_symm (Hermitian path)" if args.no_ket
                     else "k_split_reg<14, 5, ROWS> row passes + k_transpose_conj (split-operator master equation)")
            wl = f"BASELINE configs[2]: 14-atom triangular register, dephasing mesolve (rho = 4.29 GB), {args.slice_ns} ns slice at t = 1 us"
        else:
            shape = {20: (4, 5), 22: (2, 11), 24: (4, 6)}[args.atoms]
            eng = Engine.from_problems([rect_problem(*shape)], mode="sesolve")
            t0, t1, nb = 1.0, 1.0 + 1e-3 * args.slice_ns, args.atoms
            kname = KSPLIT_NAME if args.method in ("auto", "split") else KAPPLY_NAME
            wl = (f"BASELINE configs[4]: {args.atoms}-atom {shape[0]}x{shape[1]} register, sesolve, "
                  f"{args.slice_ns} ns slice at t = 1 us, method {args.method}")
        mopt = {"method": args.method} if args.workload == "cfg5" else {}
        state_fn = eng.new_state
        if args.workload == "cfg5":
            psi1 = eng.new_state()
            eng.evolve(psi1, 0.0, t0)  # the slice starts from the state the sequence has reached at t0
            state_fn = psi1.clone
        sec, stats, kms, kl, occ = timed_run(eng, state_fn, t0, t1, args.steps, args.warmup, dist, torch, **mopt)
        if args.workload == "cfg3" and not args.no_ket:
            roof = roofline_valu(2.0**14, 2.0**15, stats["n_applications"], KSPLITREG_FLOPS_PER_AMP_STAGE, kms, kl, kname, "cfg3:k_split_reg",
                                 algorithmic_flops_per_amp_stage=SPLIT_ALGORITHMIC_FLOPS_PER_AMP_STAGE)
        else:
            key = "cfg3:k_apply" if args.workload == "cfg3" else (
                "cfg5_24atoms:k_split" if (args.atoms == 24 and kname == KSPLIT_NAME) else
                "cfg5_22atoms:k_split" if (args.atoms == 22 and kname == KSPLIT_NAME) else None if args.atoms != 20
                else "cfg5:k_split" if kname == KSPLIT_NAME else "cfg5:k_apply")
            roof = roofline_hbm(nb, 1, stats, kms, kl, kname, key)
        out = {"metric": "sim-us/sec", "value": n_gpus * (t1 - t0) / sec, **common, "ms_per_step": sec * 1e3,
               "config": {"workload": wl, "passes_per_application": stats["passes"], "order": stats["last_order"],
                          "parallelism": f"replicas x{n_gpus} (a single state does not shard)"},
               "roofline": roof}
        eng.close()
        extras_ok = False
        args.no_cpu = True
    else:
        raise SystemExit(f"unknown workload {args.workload}")

    if extras_ok:
        also = []
        # the round-1 headline: 256 x 12-atom sequences (LDS-resident persistent kernel)
        eng = Engine(spread_tables(chain_problem(12), 256), mode="sesolve")
        sec, stats, kms, kl, _ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, 2, 1, None, torch)
        also.append({"workload": "cfg2: 256 independent 12-atom chain sequences, full 3.1 us", "value": 256 * T_SEQ_US / sec,
                     "unit": "sim-us/s", "ms_per_batch": sec * 1e3, "applications_per_sequence": stats["n_applications"],
                     "roofline": split_reg_roofline(12, 256, stats, kms, kl, "cfg2:k_split_reg") if ran_split_reg(stats) else
                                 roofline_valu(4096.0, 256, stats["n_applications"], KTRAJ_FLOPS_PER_AMP_STAGE, kms, kl,
                                               "k_traj<12,1024,1> (persistent, LDS-resident)", "cfg2:k_traj")})
        # ... and the same batch on the polynomial persistent kernel (the round-1 - 3 kernel of this line)
        eng.set_path(False, no_split14=True)
        sec, stats, kms, kl, _ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, 2, 1, None, torch)
        also[-1]["k_traj"] = {"value": 256 * T_SEQ_US / sec, "unit": "sim-us/s", "applications_per_sequence": stats["n_applications"],
                              "roofline": roofline_valu(4096.0, 256, stats["n_applications"], KTRAJ_FLOPS_PER_AMP_STAGE, kms, kl,
                                                        "k_traj<12,1024,1> (persistent, LDS-resident)", "cfg2:k_traj")}
        eng.close()
        # the same with per-atom complex drives (MODEL 0 of the persistent kernel): local addressing
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import local_problem

        # round 4: the phase of a drive rides on the D factors of the split-operator stages (SplitRun.gauge), so these run
        # on the real tan-form kernel by default; the polynomial kernels of rounds 1 - 3 beside it (set_path(no_split14))
        for n_c, poly_name, poly_flops in ((12, "k_traj<12,1024,0> (persistent, per-atom complex coefficients)", None),
                                           (14, "k_ket<14, KET_GAUGE> (complex drives gauged away: 4-point Gauss moments of |c| "
                                                "and d/dt arg c per step and atom)", KKET_FLOPS_PER_AMP_STAGE)):
            eng = Engine.from_problems([local_problem(n_c, seed=s, duration=401) for s in range(8)] * 32, mode="sesolve")
            sec, stats, kms, kl, _ = timed_run(eng, eng.new_state, 0.0, 0.4, 2, 1, None, torch)
            leg = {"workload": f"256 x {n_c}-atom sequences with per-atom complex, time-dependent drives (local addressing), 400 ns",
                   "value": 256 * 0.4 / sec, "unit": "sim-us/s", "ms_per_batch": sec * 1e3,
                   "stages_per_sequence": stats["n_applications"], "launches": stats["n_launches"],
                   "kernel": (KSPLITREG_NAME.replace("14, 5", f"{n_c}, {KSPLITREG_SHAPES[n_c][0]}") + "; the drive phases carried by the D factors, "
                              "4th-order 6-stage composition with one-knot sub-steps (modulated drives: nothing to merge)")
                             if ran_split_reg(stats) else poly_name}
            if ran_split_reg(stats):
                leg["roofline"] = split_reg_roofline(n_c, 256, stats, kms, kl, None)
            eng.set_path(False, no_split14=True)
            sec, stats, kms, kl, _ = timed_run(eng, eng.new_state, 0.0, 0.4, 2, 1, None, torch)
            leg["polynomial_kernel"] = {"value": 256 * 0.4 / sec, "unit": "sim-us/s", "stages_per_sequence": stats["n_applications"],
                                        "launches": stats["n_launches"], "kernel": poly_name}
            if poly_flops:
                leg["polynomial_kernel"]["roofline"] = roofline_valu(2.0**n_c, 256, stats["n_applications"], poly_flops, kms, kl,
                                                                     "k_ket<14, KET_GAUGE>")
            also.append(leg)
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
        del psi, rho
        # cfg4 end to end: 1024 noise trajectories through run_ensemble (BASELINE configs[3]); the --workload cfg4 line
        # of an N-GPU run shards the same ensemble
        c4 = cfg4_line(1024, 5, 1, None, torch, 1, {})
        also.append({"workload": "cfg4: 1024 noise trajectories of the 12-atom anneal sequence, END TO END (noise draws, "
                                 "factored lowering, solve, reference-order sampling) through run_ensemble",
                     "value": c4["value"], "unit": "trajectories/s", "ms_per_ensemble": c4["ms_per_step"],
                     "sim_us_per_s": c4["config"]["sim_us_per_s"],
                     "with_density_matrix": c4["config"]["with_density_matrix"],
                     "histogram_total": c4["config"]["histogram_total"], "n_measures": c4["config"]["n_measures"],
                     "repetitions": 5, "ms_min": c4["config"]["ms_min"], "ms_median": c4["config"]["ms_median"],
                     "ms_max": c4["config"]["ms_max"], "spread": c4["config"]["spread"]})
        # cfg5: 20 atoms over the FULL 3.1 us: split-operator passes (default) with the Lanczos exponential (the solver
        # configs[4] names) and CF4 + Taylor beside it; 24 atoms = the HBM-bound regime (40 ns slice at t = 1 us)
        eng = Engine.from_problems([rect_problem(4, 5)], mode="sesolve")
        sec, stats, kms, kl, occ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, 1, 1, None, torch)
        leg = {"workload": "cfg5: 20-atom 4x5 register, sesolve, FULL 3.1 us (default: split-operator passes)",
               "value": T_SEQ_US / sec, "unit": "sim-us/s", "seconds": sec, "stages": stats["n_applications"],
               "passes_per_stage": stats["passes"], "local_error_estimate": stats["reserved"][0],
               "norm": float(occ[-1].item()),
               "roofline": roofline_hbm(20, 1, stats, kms, kl, KSPLIT_NAME, "cfg5:k_split")}
        psi_split = timed_run.last_state.clone()
        sec, stats, kms, kl, occ = timed_run(eng, eng.new_state, 0.0, T_SEQ_US, 1, 0, None, torch, method="krylov")
        leg["krylov"] = {"workload": "the same full 3.1 us on the Lanczos exponential (ryd_opts.method = 1)",
                         "value": T_SEQ_US / sec, "unit": "sim-us/s", "seconds": sec,
                         "applications": stats["n_applications"], "krylov_dimension": stats["last_order"],
                         "max_abs_vs_split": float((timed_run.last_state - psi_split).abs().max().item()),
                         "roofline": roofline_hbm(20, 1, stats, kms, kl, "k_apply<sesolve> inside the Lanczos process "
                                                                        "(its epilogue reduces the inner products; + k_kry_update_fused: two launches per iteration)", None)}
        psi1 = eng.new_state()
        eng.evolve(psi1, 0.0, 1.0)
        sec, stats, kms, kl, occ = timed_run(eng, psi1.clone, 1.0, 1.02, 2, 1, None, torch, method="taylor")
        leg["taylor"] = {"workload": "20 ns slice at t = 1 us on CF4 + Taylor", "value": 0.02 / sec, "unit": "sim-us/s",
                         "taylor_order": stats["last_order"],
                         "roofline": roofline_hbm(20, 1, stats, kms, kl, KAPPLY_NAME, "cfg5:k_apply")}
        also.append(leg)
        del psi1, psi_split
        eng.close()
        eng = Engine.from_problems([rect_problem(4, 6)], mode="sesolve")
        psi1 = eng.new_state()
        eng.evolve(psi1, 0.0, 1.0)  # the slice starts from the state the sequence has reached at 1 us
        sec, stats, kms, kl, occ = timed_run(eng, psi1.clone, 1.0, 1.04, 2, 1, None, torch)
        also.append({"workload": "cfg5: 24-atom 4x6 register, sesolve, 40 ns slice at t = 1 us",
                     "value": 0.04 / sec, "unit": "sim-us/s", "stages": stats["n_applications"],
                     "passes_per_stage": stats["passes"], "local_error_estimate": stats["reserved"][0],
                     "roofline": roofline_hbm(24, 1, stats, kms, kl, KSPLIT_NAME, "cfg5_24atoms:k_split")})
        del psi1
        eng.close()
        # the multi-level ("all" basis, 3 levels) and XY path, k_gen_apply_fused behind ryd_solve of a
        # GeneralEngine (tools/general_bench.py; kernel times: profiles/r06_general_path.md).  The 10-atom and the 12-atom
        # legs first: they are the ones the contract line quotes
        for extra_path in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
            if extra_path not in sys.path:
                sys.path.insert(0, extra_path)
        import general_bench

        gen_legs = general_bench.legs(("fused",))
        gen_legs.sort(key=lambda leg: -int(leg["dim"]) if leg["workload"].startswith("f-1:") else -int(leg["n_atoms"]))
        also.extend(gen_legs)
        out["also"] = also
        bw = device_copy_bandwidth(torch)
        out["device_copy_GBps"] = bw
        if out.get("lindblad", {}).get("roofline", {}).get("hbm_equivalent_GBps"):
            out["lindblad"]["hbm_equivalent_over_copy_bandwidth"] = out["lindblad"]["roofline"]["hbm_equivalent_GBps"] / bw

    if rank == 0:
        out["cpu_baseline"] = cpu_result
        if collective is not None:
            out["collective"] = collective

    if rank == 0:
        # everything measured goes to bench_detail.json and to an EARLIER stdout line (prefix DETAIL_PREFIX, so that
        # no parser takes it for the contract line); the LAST line is the short contract line the driver parses
        detail = json.dumps(out)
        try:
            with open(os.environ.get("RYD_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")), "w") as fh:
                fh.write(detail + "\n")
        except OSError as exc:  # a read-only checkout still prints both lines
            print(f"bench.py: bench_detail.json not written ({exc})", file=sys.stderr)
        print(DETAIL_PREFIX + detail, flush=True)
        line = json.dumps(driver_line(out))
        assert len(line) <= MAX_LINE_BYTES, len(line)
        print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
