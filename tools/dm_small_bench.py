"""Small density matrices: the persistent kernel vs the tiled kernels (dev probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import chain_problem
from pulser_amd.engine import Engine

ops = [(np.sqrt(0.1), "sigma_rr"), (np.sqrt(0.05), "sigma_gr")]
for n, B in ((2, 1), (4, 1), (6, 1), (4, 256), (6, 256)):
    for force in (False, True):
        eng = Engine.from_problems([chain_problem(n, collapse_ops=ops)] * B, mode="mesolve")
        eng.set_path(force)
        st = eng.new_state(); eng.evolve(st, 0.0, 0.01); torch.cuda.synchronize(); eng.reset_stats()
        st = eng.new_state()
        t1 = 3.1 if not force or B == 1 else 0.31
        t0 = time.time(); eng.evolve(st, 0.0, t1); torch.cuda.synchronize(); dt = time.time() - t0
        s = eng.stats()
        print(f"N={n} B={B} {'tiled' if force else 'persistent'}: {t1*B/dt:.1f} sim-us/s "
              f"({dt/(t1/3.1)*1e3:.1f} ms per 3.1-us batch), launches {s['n_launches']}", flush=True)
        eng.close()
