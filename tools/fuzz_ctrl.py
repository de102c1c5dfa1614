#!/usr/bin/env python
"""Fuzz of the split-operator step-size controller: seeded random sequences from Pulser's waveform
families (tests/helpers.py: fuzz_case), default path against CF4 + Taylor at tol 1e-12; prints error, booked estimate,
their ratio, worst first.  python tools/fuzz_ctrl.py [first_seed] [n_cases]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import fuzz_case
from pulser_amd.engine import Engine

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
t_all = time.perf_counter()
for seed in range(first, first + count):
    probs, desc = fuzz_case(seed)
    t_end = (probs[0]["duration"] - 1) * 1e-3
    tic = time.perf_counter()
    with Engine.from_problems(probs, mode="sesolve") as eng:
        ref = eng.new_state()
        eng.evolve(ref, 0.0, t_end, method="taylor", tol=1e-12, magnus_tol=1e-12)
        st = eng.new_state()
        eng.reset_stats()
        import warnings
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            eng.evolve(st, 0.0, t_end)
        s = eng.stats()
        err = float((st - ref).abs().max())
        err2 = float(torch.linalg.vector_norm(st - ref, dim=1).max())  # largest 2-norm of a sequence's error vector
    est = s["reserved"][0]
    rows.append((err / max(est, 1e-300) if est > 0 else 0.0, err, est, s["n_applications"], s["n_launches"], s["reserved"][3],
                 time.perf_counter() - tic, len(w), desc))
    print(f"err {err:.2e} est {est:.2e} ratio {rows[-1][0]:6.2f} err2 {err2:.2e} stages {s['n_applications']:6d} launches {s['n_launches']:5d} "
          f"rollbacks {s['reserved'][3]:.0f} warn {len(w)} {rows[-1][6]:.2f}s  {desc}", flush=True)
print(f"\n{count} cases in {time.perf_counter() - t_all:.1f} s; worst error {max(r[1] for r in rows):.2e}; "
      f"worst error / estimate {max(r[0] for r in rows):.2f}")
for r in sorted(rows, reverse=True)[:8]:
    print(f"  ratio {r[0]:6.2f} err {r[1]:.2e} est {r[2]:.2e}  {r[8]}")
bad = [r for r in rows if r[1] > 1e-7 or (r[2] > 0 and r[1] > max(4 * r[2], 2e-9))]
print(f"violations (err > 1e-7 or err > max(4 est, 2e-9)): {len(bad)}")
