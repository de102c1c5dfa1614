#!/usr/bin/env python
"""Where along a fuzz case does the default path leave the reference?  python tools/fuzz_locate.py SEED [step_ns]
(error against CF4 + Taylor at tol 1e-13 at every step_ns; RYD_DEV=1 RYD_SPLIT_TRACE=1 adds the controller's checks)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import fuzz_case
from pulser_amd.engine import Engine
seed = int(sys.argv[1]); step = int(sys.argv[2]) if len(sys.argv) > 2 else 16
probs, desc = fuzz_case(seed)
print(desc)
T = probs[0]["duration"] - 1
times = np.unique(np.concatenate([np.arange(0, T, step), [T]])) * 1e-3
a = probs[0]["samples"]["Global"]["ground-rydberg"]
with Engine.from_problems(probs, mode="sesolve") as eng:
    ref = eng.solve(eng.new_state(), times, method="taylor", tol=1e-13, magnus_tol=1e-13)
    out = eng.solve(eng.new_state(), times)
    s = eng.stats()
err = (out - ref).abs().amax(dim=(1, 2)).cpu().numpy()
print("estimate", s["reserved"][0], "stages", s["n_applications"])
for k in range(len(err)):
    i = int(round(times[k + 1] * 1e3))
    print(f"t = {times[k+1]*1e3:6.0f} ns  err {err[k]:.2e}   amp {a['amp'][min(i, T)]:7.2f} det {a['det'][min(i, T)]:8.2f}")
