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


def three_level_problem(n, T=201, seed=5):
    from pulser_amd import problem as P

    rng = np.random.default_rng(seed)
    lay = P.square_rect(3, 3) if n == 9 else P.square_rect(2, n // 2)
    coords = P.register_coords(lay, 6.5)
    t = np.arange(T) / 1000.0
    prob = P.make_ising_problem(coords, {"amp": 6.0 + 2.0 * np.sin(40 * t), "det": -3.0 + 50 * t, "phase": 0.4 * np.ones(T)})
    prob["eigenbasis"] = ["r", "g", "h"]
    prob["basis_name"] = "all"
    prob["samples"]["Local"] = {"digital": {q: {"amp": rng.uniform(2, 8) * np.ones(T), "det": rng.uniform(-3, 3) * np.ones(T),
                                                 "phase": rng.uniform(0, 1) * np.ones(T)} for q in (0, n // 2, n - 2)}}
    init = np.zeros(3**n, dtype=complex)
    init[sum(1 * 3**k for k in range(n))] = 1.0  # |g...g>
    return prob, init, (T - 1) * 1e-3


def xy_problem(n=12):
    """The reference's mesolve-XY test sequence (tests/golden/noisy_xy_2.npz: inputs captured from pulser-core) on a 12-atom
    2 x 6 register at the same 4-um pitch: global XY channel, magnetic field (0, 0, 30)."""
    from helpers import load_fixture
    from pulser_amd import QutipEmulator, problem as P
    from pulser_amd.hamiltonian_data import SequenceInputs

    prob, _ = load_fixture("noisy_xy_2.npz")
    inp = dict(prob["inputs"])
    inp["coords"] = P.register_coords(P.square_rect(2, n // 2), 4.0)
    inp["qubit_ids"] = tuple(f"atom{k}" for k in range(n))
    ch = dict(inp["channels"][0])
    ch["slots"] = [np.array([int(s[0]), int(s[1])] + list(range(n)), dtype=np.int64) for s in ch["slots"]]
    inp["channels"] = [ch]
    emu = QutipEmulator(SequenceInputs.from_dict(inp), sampling_rate=1.0)
    p = emu._current_problem
    return p, np.asarray(emu.initial_state).reshape(-1), (int(p["duration"]) - 1) * 1e-3


def run_leg(label, prob, init, t_end, mesolve=False, multi=None):
    import torch

    from pulser_amd.engine import GeneralEngine
    from pulser_amd.general import lower_general

    d, n = len(prob["eigenbasis"]), prob["n_qudits"]
    tic = time.perf_counter()
    tables = lower_general(prob, mesolve=mesolve, matrix_free=True)
    lower_s = time.perf_counter() - tic
    with GeneralEngine(tables) as eng:
        if multi is not None:
            eng.set_path(bool(multi))
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
            "roofline": {"bound": "hbm", "kernel": "k_gen_apply_sites", "achieved": bytes_per_app * apps / best / 1e9,
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bytes_per_app * apps / best / HBM_PEAK,
                         "algorithmic_bytes_per_launch": bytes_per_app, "traffic": None,
                         "basis": "wall clock of the solve (launch-bound at these sizes: d^N x 32 B = "
                                  f"{bytes_per_app / 1e6:.2f} MB per application)"}}


def legs():
    out = []
    for n in (9, 10):
        prob, init, t_end = three_level_problem(n)
        out.append(run_leg(f"f-1: 3-level 'all' basis, {n} atoms (3^{n} amplitudes), global ground-rydberg + 3 local raman "
                           f"drives, sesolve, {t_end * 1e3:.0f} ns", prob, init, t_end))
    # XY: a SLICE (the exchange couplings at 4 um are 567 rad/us: 670 000 generator applications over the full microsecond -
    # round 6 measured 8 minutes for it on the one-workgroup kernel before the selection rule below existed)
    slice_us = float(os.environ.get("GEN_XY_SLICE_US", "0.02"))
    for n in (8, 12):
        prob, init, _ = xy_problem(n)
        for multi, tag in ((None, "default path"), (True, "multi-launch forced")):
            out.append(run_leg(f"f-4: XY exchange, {n} atoms (2 x {n // 2} at 4 um), global XY channel, sesolve, "
                               f"{slice_us * 1e3:.0f} ns slice [{tag}]", prob, init, slice_us, multi=multi))
    return out


if __name__ == "__main__":
    res = legs()
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        for r in res:
            print(f"{r['workload']}\n   {r['value']:.3f} sim-us/s ({r['seconds'] * 1e3:.1f} ms), {r['applications']} applications in "
                  f"{r['launches']} launches, {r['n_terms']} terms, {r['us_per_application_wall']:.2f} us / application (wall), Taylor order {r['taylor_order']}, "
                  f"HBM frac {r['roofline']['frac']:.4f} ({r['roofline']['achieved']:.1f} GB/s), norm {r['norm']:.12f}")
