#!/usr/bin/env python
"""Measured figures of the multi-level / XY path (VERDICT r05 "missing" 4).

    python tools/general_bench.py [--json]

Legs (sesolve, matrix-free site-fused terms: k_gen_apply_sites behind ryd_solve of a GeneralEngine):
  f-1  3-level "all" basis (ground-rydberg global + raman local channels, hamiltonian.py:145-200, 276-294) at 9 and 10
       atoms (3^9 = 19 683, 3^10 = 59 049 amplitudes);
  f-4  XY (exchange interaction, hamiltonian_data.py:913-931) at 12 atoms (4 096 amplitudes).
Per leg: sim-us/s, generator applications, wall time per application, and the HBM roofline of the application kernel with
ALGORITHMIC bytes = 32 B x d^N per application (read + write of the complex128 vector).  Kernel time: rocprofv3
--kernel-trace --stats of this script (tools/profile_r06.sh -> profiles/r06_general_kernel_stats.csv)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK = 8.0e12


from helpers import three_level_problem, xy_problem  # noqa: E402  (tests/helpers.py: the GPU tests use the same problems)


def run_leg(label, prob, init, t_end, mesolve=False, multi=None, variant="fused"):
    import torch

    from pulser_amd.engine import GeneralEngine
    from pulser_amd.general import lower_general

    d, n = len(prob["eigenbasis"]), prob["n_qudits"]
    tic = time.perf_counter()
    tables = lower_general(prob, mesolve=mesolve, matrix_free=True)
    lower_s = time.perf_counter() - tic
    with GeneralEngine(tables) as eng:
        # variant "sites": the round-3 kernel (k_gen_apply_sites) instead of the padded site tables (k_gen_apply_fused)
        if multi is not None or variant != "fused":
            eng.set_path(bool(multi), no_fused=variant == "sites")
        eng.solve(eng.new_state(init), [0.0, min(0.002, t_end)])  # warm-up: code objects, work buffers
        torch.cuda.synchronize()
        best, st, out = np.inf, None, None
        for _ in range(3):
            eng.reset_stats() if hasattr(eng, "reset_stats") else None
            s0 = eng.stats()
            psi = eng.new_state(init)
            torch.cuda.synchronize()
            tic = time.perf_counter()
            out = eng.solve(psi, [0.0, t_end])
            torch.cuda.synchronize()
            dt = time.perf_counter() - tic
            s1 = eng.stats()
            if dt < best:
                best = dt
                st = {k: s1[k] - s0[k] for k in ("n_applications", "n_launches", "n_steps")}
                st["last_order"] = s1["last_order"]
        norm = float(torch.linalg.vector_norm(out[-1]).item())
    dim = d**n
    apps = max(st["n_applications"], 1)
    bytes_per_app = 32.0 * dim
    return {"workload": label, "levels": d, "n_atoms": n, "dim": dim, "n_terms": int(len(tables.series)), "sim_us": t_end, "value": t_end / best, "unit": "sim-us/s",
            "seconds": best, "lowering_s": lower_s, "applications": st["n_applications"], "launches": st["n_launches"],
            "steps": st["n_steps"], "taylor_order": st["last_order"], "us_per_application_wall": best * 1e6 / apps,
            "norm": norm,
            "variant": variant,
            "roofline": {"bound": "hbm", "kernel": "k_gen_apply_fused" if variant == "fused" else "k_gen_apply_sites", "achieved": bytes_per_app * apps / best / 1e9,
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bytes_per_app * apps / best / HBM_PEAK,
                         "algorithmic_bytes_per_launch": bytes_per_app, "traffic": None,
                         "basis": "wall clock of the solve (launch-bound at these sizes: d^N x 32 B = "
                                  f"{bytes_per_app / 1e6:.2f} MB per application)"}}


def legs(variants=("fused", "sites")):
    out = []
    for n in (9, 10):
        prob, init, t_end = three_level_problem(n)
        for v in variants:
            out.append(run_leg(f"f-1: 3-level 'all' basis, {n} atoms (3^{n} amplitudes), global ground-rydberg + 3 local raman "
                               f"drives, sesolve, {t_end * 1e3:.0f} ns [{v}]", prob, init, t_end, variant=v))
    # XY: a SLICE (the exchange couplings at 4 um are 567 rad/us: hundreds of thousands of generator applications over the
    # full microsecond)
    slice_us = float(os.environ.get("GEN_XY_SLICE_US", "0.02"))
    for n in (8, 12):
        prob, init, _ = xy_problem(n)
        for v in variants:
            out.append(run_leg(f"f-4: XY exchange, {n} atoms (2 x {n // 2} at 4 um), global XY channel, sesolve, "
                               f"{slice_us * 1e3:.0f} ns slice [{v}, multi-launch]", prob, init, slice_us, multi=True, variant=v))
        if n == 8:
            out.append(run_leg(f"f-4: XY exchange, {n} atoms, the same slice [default path selection]", prob, init, slice_us))
    return out


if __name__ == "__main__":
    res = legs(("fused",) if "--fused-only" in sys.argv else ("fused", "sites"))
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        for r in res:
            print(f"{r['workload']}\n   {r['value']:.3f} sim-us/s ({r['seconds'] * 1e3:.1f} ms), {r['applications']} applications in "
                  f"{r['launches']} launches, {r['n_terms']} terms, {r['us_per_application_wall']:.2f} us / application (wall), Taylor order {r['taylor_order']}, "
                  f"HBM frac {r['roofline']['frac']:.4f} ({r['roofline']['achieved']:.1f} GB/s), norm {r['norm']:.12f}")
