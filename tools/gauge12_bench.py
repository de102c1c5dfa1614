"""Dev probe: 256 x N-atom sequences with per-atom complex drives: LDS-resident k_traj (MODEL 0, complex coefficients)
against the register-resident k_ket in gauge mode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import local_problem
from pulser_amd.engine import Engine

for n in (10, 11, 12, 13):
    probs = [local_problem(n, seed=s, duration=401) for s in range(8)] * 32
    for force in (False, True):
        with Engine.from_problems(probs, mode="sesolve") as eng:
            eng.set_path(False, force_ket=force)
            st = eng.new_state(); eng.evolve(st, 0.0, 0.02); torch.cuda.synchronize()
            st = eng.new_state(); eng.reset_stats(); torch.cuda.synchronize(); t0 = time.time()
            eng.evolve(st, 0.0, 0.4); torch.cuda.synchronize(); dt = time.time() - t0
            s = eng.stats()
            print(f"N={n} force_ket={force}: {256 * 0.4 / dt:.0f} sim-us/s ({dt * 1e3:.1f} ms), launches {s['n_launches']}, stages {s['n_applications']}", flush=True)
