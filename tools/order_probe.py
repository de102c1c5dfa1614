"""How conservative is the a-priori Taylor order? Fixed orders / looser tolerances vs the
tight oracle states of the 12-atom anneal fixture (dev probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import load_fixture, with_anneal_samples
from pulser_amd.engine import Engine

prob, extra = load_fixture("cfg2_chain12_anneal.npz")
prob = with_anneal_samples(prob)
times = np.asarray(extra["eval_times"], float)
ref = np.asarray(extra["oracle_states_tight"])
for kw in ({}, {"tol": 1e-10}, {"tol": 1e-9}, {"tol": 1e-8}, {"tol": 1e-7},
           {"taylor_order": 8}, {"taylor_order": 7}, {"taylor_order": 6}, {"taylor_order": 5},
           {"taylor_order": 4}, {"magnus_tol": 1e-9}, {"magnus_tol": 1e-8}, {"tol": 1e-9, "magnus_tol": 1e-9}):
    with Engine.from_problems([prob], mode="sesolve") as eng:
        st = eng.new_state()
        worst = 0.0
        for i in range(1, len(times)):
            eng.evolve(st, times[i - 1], times[i], **kw)
            worst = max(worst, float(np.max(np.abs(st.cpu().numpy()[0] - ref[i]))))
        s = eng.stats()
        nrm = float(np.linalg.norm(st.cpu().numpy()[0]))
        print(f"{kw}: applications {s['n_applications']}, steps {s['n_steps']}, max err vs tight oracle {worst:.2e}, |norm-1| {abs(nrm-1):.1e}", flush=True)
