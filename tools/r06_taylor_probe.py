#!/usr/bin/env python
"""round 6: CF4 + Taylor (tol = magnus_tol = 1e-12) against the tight oracle on the strongly interacting fuzz cases, by max_step:
does the error fall like n^-4 (a 4th-order Magnus error the a-priori estimate does not see)?  python tools/r06_taylor_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import fuzz_case
from pulser_amd.engine import Engine
fx = np.load(os.path.join(ROOT, "tests", "golden", "fuzz_oracle_holdout.npz"))
for k, seed in enumerate(fx["seeds"]):
    if int(seed) not in (2799, 2745, 2013, 2343, 2685):
        continue
    probs, desc = fuzz_case(int(seed))
    row = [r for r in range(len(fx["state_owner"])) if fx["state_owner"][r][0] == k][0]
    ref = torch.from_numpy(fx["states"][row][: 2 ** int(fx["state_atoms"][row])]).cuda()
    t_end = (probs[0]["duration"] - 1) * 1e-3
    s = probs[0]["samples"]["Global"]["ground-rydberg"]
    c = 0.5 * s["amp"] * np.exp(-1j * s["phase"])
    print(desc, "| max |c'| =", f"{np.abs(np.diff(c)).max() * 1e3:.0f} rad/us^2, max |c| = {np.abs(c).max():.1f}")
    with Engine.from_problems(probs[:1], mode="sesolve") as eng:
        for ms in (0.0, 0.5e-3, 0.25e-3, 0.125e-3):
            st = eng.new_state(); eng.reset_stats()
            eng.evolve(st, 0.0, t_end, method="taylor", tol=1e-12, magnus_tol=1e-12, max_step=ms)
            print(f"   max_step {ms * 1e3:5.3f} ns: |taylor - oracle| = {float((st[0] - ref).abs().max()):.2e}  applications {eng.stats()['n_applications']}")
