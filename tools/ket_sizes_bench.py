"""Dev probe: where the register-resident ket kernel / split-operator rows pay at 12 and 13 atoms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import chain_problem
from pulser_amd.engine import Engine

def run(eng, t0, t1):
    st = eng.new_state(); torch.cuda.synchronize(); eng.reset_stats()
    tic = time.perf_counter(); eng.evolve(st, t0, t1); torch.cuda.synchronize()
    return time.perf_counter() - tic, eng.stats()

for n in (12, 13):
    for force in (False, True):
        with Engine.from_problems([chain_problem(n)] * 256, mode="sesolve") as eng:
            eng.set_path(False, force_ket=force)
            run(eng, 0.0, 0.01)
            dt, s = run(eng, 1.0, 1.1)
            print(f"sesolve 256 x {n} atoms, force_ket={force}: {256*0.1/dt:.0f} sim-us/s, stats {s}", flush=True)
ops = [(float(np.sqrt(0.1)), "sigma_rr")]
for n in (10, 11, 12, 13):
    for rows in (True, False):
        with Engine.from_problems([chain_problem(n, ops)], mode="mesolve") as eng:
            eng.set_path(False, force_ket=rows, no_ket=not rows)
            run(eng, 1.0, 1.004)
            span = 0.016 if n <= 12 else 0.008
            dt, s = run(eng, 1.0, 1.0 + span)
            print(f"mesolve {n} atoms, rows={rows}: {dt/span/1e3*1e3:.3f} ms per sim-ns, stats {s}", flush=True)
