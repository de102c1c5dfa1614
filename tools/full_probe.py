#!/usr/bin/env python
"""Engine-level probe: a solve with evaluation times at every knot / every 10th knot / the end points only, on the
default path and on k_ket (set_path(force_ket=True)).  python tools/full_probe.py [atoms]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import tri_problem, chain_problem
from pulser_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
prob = tri_problem(2, 7) if n == 14 else chain_problem(n)
grid = np.arange(3101) * 1e-3
for label, times in (("minimal", grid[[0, -1]]), ("every10", grid[::10]), ("full", grid)):
    for path in ("default", "force_ket"):
        eng = Engine.from_problems([prob], mode="sesolve")
        if path == "force_ket":
            eng.set_path(False, force_ket=True)
        ref = None
        for rep in range(2):
            st = eng.new_state()
            eng.reset_stats()
            torch.cuda.synchronize()
            tic = time.perf_counter()
            out = eng.solve(st, times, store=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - tic
        s = eng.stats()
        print(f"[probe] n={n} {label:8} {path:10} {dt*1e3:8.1f} ms stages={s['n_applications']} launches={s['n_launches']} "
              f"est={s['reserved'][0]:.1e} norm={float((st.abs()**2).sum()):.12f}", flush=True)
        if path == "default":
            keep = out[-1].clone()
        else:
            print(f"        max|default - force_ket| final = {float((out[-1]-keep).abs().max()):.2e}")
        eng.close()
