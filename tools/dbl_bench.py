"""mesolve with a double-flip dissipator (relaxation): the pair-pass tiling (dev probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import chain_problem
from pulser_amd.engine import Engine

for n, t1 in ((8, 0.05), (10, 0.05), (12, 0.012), (13, 0.004), (14, 0.003)):
    for ops, tag in (([(np.sqrt(0.1), "sigma_rr")], "dephasing"), ([(np.sqrt(0.1), "sigma_gr")], "relaxation")):
        eng = Engine.from_problems([chain_problem(n, collapse_ops=ops)], mode="mesolve")
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.002)
        torch.cuda.synchronize(); eng.reset_stats()
        t0 = time.time(); eng.evolve(st, 0.002, t1); torch.cuda.synchronize(); dt = time.time() - t0
        s = eng.stats()
        print(f"N={n} {tag}: {(t1-0.002)/dt:.4f} sim-us/s; passes {s['passes']} apps {s['n_applications']} "
              f"launches {s['n_launches']}; {dt/s['n_applications']*1e3:.3f} ms/application; "
              f"alg BW {32.0*4**n*s['n_applications']/dt/1e12:.2f} TB/s", flush=True)
        eng.close()
