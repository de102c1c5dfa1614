"""Dev probe: split-operator passes with 2^12 tiles (k_split12, two passes per stage from 21 atoms) against the
larger tiles of k_split_t (2^13: 21-22 atoms, 2^14: 23-25 atoms; one pass per stage).  Fixed sub-steps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

SHAPES = {21: (3, 7), 22: (2, 11), 23: (1, 23), 24: (4, 6), 25: (5, 5)}
for n in [int(v) for v in sys.argv[1:]] or [22, 24]:
    coords = P.register_coords(P.square_rect(*SHAPES[n]), blockade_radius())
    prob = P.make_ising_problem(coords, P.anneal_samples())
    res = {}
    for small in (True, False):
        with Engine.from_problems([prob], mode="sesolve") as eng:
            eng.set_path(False, split_fixed=True, split_small_tiles=small)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.004, method="split")
            ref = st.clone()
            eng.reset_stats()
            torch.cuda.synchronize(); t0 = time.time()
            eng.evolve(st, 0.004, 0.014, method="split")
            torch.cuda.synchronize(); dt = time.time() - t0
            s = eng.stats()
            res[small] = st.cpu().numpy()
            print(f"N={n} tiles {'2^12' if small else 'large'}: 10 ns in {dt*1e3:.1f} ms; launches {s['n_launches']}, stages {s['n_applications']}, "
                  f"{dt / s['n_applications'] * 1e6:.1f} us per stage, passes/stage {s['passes']}", flush=True)
    print(f"N={n} max |large - 2^12| = {np.max(np.abs(res[True] - res[False])):.2e}; norm {np.linalg.norm(res[False]):.12f}", flush=True)
