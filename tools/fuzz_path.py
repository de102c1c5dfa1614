#!/usr/bin/env python
"""Error of the default path along a fuzz case WITHOUT evaluation times in the way: one cold solve 0 -> t per t (the schedule,
the allowances - the budget is per pulse sequence - and the checks before t are those of the whole solve; tools/fuzz_locate.py
asks for intermediate states, and evaluation times cut the sub-steps).  python tools/fuzz_path.py SEED [step_ns]"""
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
a = probs[0]["samples"]["Global"]["ground-rydberg"]
with Engine.from_problems(probs, mode="sesolve") as eng:
    for t in list(range(step, T, step)) + [T]:
        ref = eng.new_state(); eng.evolve(ref, 0.0, t * 1e-3, method="taylor", tol=1e-13, magnus_tol=1e-13)
        st = eng.new_state(); eng.reset_stats(); eng.evolve(st, 0.0, t * 1e-3); s = eng.stats()
        print(f"t = {t:5d} ns  err {float((st - ref).abs().max()):.2e}  est {s['reserved'][0]:.2e} stages {s['n_applications']:5d} "
              f"rollbacks {s['reserved'][3]:.0f}  amp {a['amp'][min(t, T)]:7.2f} det {a['det'][min(t, T)]:8.2f}", flush=True)
