#!/usr/bin/env python
"""RYD_DEV=1 RYD_SPLIT_TRACE=1 python tools/trace_ctrl.py [atoms] [minimal|every10|full]: the controller's checks of one solve."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import tri_problem, chain_problem
from pulser_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
mode = sys.argv[2] if len(sys.argv) > 2 else "minimal"
prob = tri_problem(2, 7) if n == 14 else chain_problem(n)
grid = np.arange(3101) * 1e-3
times = {"minimal": grid[[0, -1]], "every10": grid[::10], "full": grid}[mode]
eng = Engine.from_problems([prob], mode="sesolve")
st = eng.new_state()
eng.solve(st, times, store=True)
torch.cuda.synchronize()
print(eng.stats())
