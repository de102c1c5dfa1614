#!/usr/bin/env python
"""Snapshots stored inside the runs of k_split_reg (k_split_reg<.., SNAP> + k_split_snap_close) against a closed run per
evaluation time (set_path(snaps_outside=True)) and against k_ket: every stored time.  python tools/snap_check.py [atoms]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bench import tri_problem, chain_problem
from pulser_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
prob = tri_problem(2, 7) if n == 14 else chain_problem(n)
grid = np.arange(3101) * 1e-3
for label, times in (("minimal", grid[[0, -1]]), ("every10", grid[::10]), ("full", grid), ("ragged", np.unique(np.concatenate([grid[::37], grid[5:40], [0.4005, 1.23456, 3.1]])))):
    outs = {}
    for path in ("inside", "outside", "taylor"):
        eng = Engine.from_problems([prob] * 3, mode="sesolve")
        kw = {}
        if path == "outside":
            eng.set_path(False, snaps_outside=True)
        if path == "taylor":
            kw = dict(method="taylor", tol=1e-12)
        for rep in range(2):
            st = eng.new_state()
            eng.reset_stats()
            torch.cuda.synchronize()
            tic = time.perf_counter()
            out = eng.solve(st, times, store=True, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - tic
        s = eng.stats()
        outs[path] = out
        print(f"[snap] n={n} {label:8} {path:8} {dt*1e3:8.1f} ms stages={s['n_applications']} launches={s['n_launches']} "
              f"est={s['reserved'][0]:.1e}", flush=True)
        eng.close()
    for a, b in (("inside", "outside"), ("inside", "taylor"), ("outside", "taylor")):
        d = (outs[a] - outs[b]).abs().amax(dim=(1, 2))
        print(f"        max over times |{a} - {b}| = {float(d.max()):.2e} (at index {int(d.argmax())}; final {float(d[-1]):.2e})")
