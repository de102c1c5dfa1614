#!/usr/bin/env python
"""One fuzz case in detail: RYD_DEV=1 RYD_SPLIT_TRACE=1 python tools/fuzz_one.py SEED  (reference convergence + the default path)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import fuzz_case
from pulser_amd.engine import Engine
seed = int(sys.argv[1])
probs, desc = fuzz_case(seed)
print(desc)
t_end = (probs[0]["duration"] - 1) * 1e-3
with Engine.from_problems(probs, mode="sesolve") as eng:
    refs = {}
    for tol, mt in ((1e-12, 1e-12), (1e-13, 1e-13)):
        r = eng.new_state(); eng.evolve(r, 0.0, t_end, method="taylor", tol=tol, magnus_tol=mt); refs[tol] = r
    print("reference convergence: |taylor(1e-12) - taylor(1e-13)| =", float((refs[1e-12] - refs[1e-13]).abs().max()))
    st = eng.new_state(); eng.reset_stats(); eng.evolve(st, 0.0, t_end); s = eng.stats()
    print("default: err vs 1e-13 ref", float((st - refs[1e-13]).abs().max()), "est", s["reserved"][0], "stages", s["n_applications"], "rollbacks", s["reserved"][3])
    for b in range(len(probs)):
        print("  seq", b, "err", float((st[b] - refs[1e-13][b]).abs().max()))
a = probs[0]["samples"]["Global"]["ground-rydberg"]
kinks = np.nonzero(np.abs(np.diff(a["amp"], 2)) > 1e-6)[0]
print("amp max", a["amp"].max(), "det range", a["det"].min(), a["det"].max())
