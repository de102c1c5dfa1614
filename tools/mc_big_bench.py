"""Quantum-jump overhead on the multi-launch kernels for large registers (dev probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import chain_problem
from pulser_amd.engine import Engine
from pulser_amd.terms import lower

ops = [(np.sqrt(2 * 0.05), "sigma_rr"), (np.sqrt(0.02), "sigma_gr")]
for n, B, t1 in ((14, 64, 0.05), (16, 64, 0.02), (18, 16, 0.02), (20, 4, 0.02), (22, 1, 0.01)):
    tables = lower([chain_problem(n, collapse_ops=ops)] * B)
    for mode in ("sesolve", "mcsolve"):
        with Engine(tables, mode=mode) as eng:
            st = eng.new_state()
            seeds = np.arange(B, dtype=np.uint64)
            run = (lambda a, b: eng.mc_solve(st, [a, b], seeds, store=False)) if mode == "mcsolve" \
                else (lambda a, b: eng.evolve(st, a, b))
            run(0.0, 0.002); torch.cuda.synchronize(); eng.reset_stats()
            t0 = time.time(); run(0.002, t1); torch.cuda.synchronize(); dt = time.time() - t0
            s = eng.stats()
            print(f"N={n} B={B} {mode}: {(t1-0.002)*B/dt:.3f} sim-us/s; launches {s['n_launches']} steps {s['n_steps']} "
                  f"apps {s['n_applications']}; {dt/s['n_steps']*1e6:.1f} us/step", flush=True)
