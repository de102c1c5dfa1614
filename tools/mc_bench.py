"""Quantum-jump throughput: 1024 trajectories of the 12-atom anneal sequence with
dephasing + relaxation, persistent kernel vs the multi-launch kernels."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pulser_amd import problem as P
from pulser_amd.engine import Engine
from pulser_amd.terms import lower

n, B = 12, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
coords = P.register_coords(P.square_rect(1, n), 8.692)
prob = P.make_ising_problem(coords, P.anneal_samples())
prob["collapse_ops"] = [(np.sqrt(2 * 0.05), "sigma_rr"), (np.sqrt(0.02), "sigma_gr")]
tables = lower([prob] * B)
seeds = np.arange(B, dtype=np.uint64) + np.uint64(1000)
for generic in (False, True):
    with Engine(tables, mode="mcsolve") as eng:
        eng.set_path(generic)
        import torch
        st = eng.new_state()
        eng.mc_solve(st, [0.0, 0.05], seeds, store=False)
        torch.cuda.synchronize()
        st = eng.new_state()
        t = time.time()
        eng.mc_solve(st, [0.0, 3.1], seeds, store=False)
        torch.cuda.synchronize()
        dt = time.time() - t
        print(f"generic={generic}: {dt:.3f} s for {B} trajectories, {B*3.1/dt:.0f} sim-us/s, "
              f"jumps/traj {eng.mc_jumps().mean():.2f}, stats {eng.stats()}", flush=True)
