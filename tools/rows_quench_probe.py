#!/usr/bin/env python
"""Which row path is right on a quench?  12-atom triangular register at R_b, dephasing 0.05/us, the all-ground matrix
dropped into the anneal at t = 0.3 us, 20 ns: split-operator rows (default), k_ket rows at the default and at a tight
tolerance, the multi-launch Lindbladian at a tight tolerance."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine
ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr")]
prob = P.make_ising_problem(P.register_coords(P.triangular_rect(2, 6), blockade_radius()), P.anneal_samples(), collapse_ops=ops)
t0, t1 = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (0.3, 0.32)
outs = {}
for name, path, opts in (("split", {}, {}), ("ket", {"rows_ket": True}, {}), ("ket_tight", {"rows_ket": True}, {"tol": 1e-13, "magnus_tol": 1e-12}),
                         ("lindbladian_tight", {"no_ket": True}, {"tol": 1e-13, "magnus_tol": 1e-12})):
    with Engine.from_problems([prob], mode="mesolve") as eng:
        eng.set_path(False, **path)
        st = eng.new_state()
        eng.evolve(st, t0, t1, **opts)
        outs[name] = st.clone()
        print(name, eng.stats()["n_applications"], eng.stats()["reserved"][:4], flush=True)
for a in outs:
    print(a, " ".join(f"{b}: {float((outs[a]-outs[b]).abs().max()):.2e}" for b in outs if b != a))
