import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from helpers import load_fixture, with_anneal_samples
from pulser_amd.engine import Engine
prob, extra = load_fixture("ns_tri14_anneal.npz"); prob = with_anneal_samples(prob)
ref = np.asarray(extra["oracle_states_tight"])[-1]
for name, kw in (("one launch per run (default)", {}), ("passes (split_no_loop)", {"split_no_loop": True})):
    with Engine.from_problems([prob], mode="sesolve") as eng:
        eng.set_path(False, **kw)
        st = eng.new_state(); eng.evolve(st, 0.0, 3.1)
        best = 1e9
        for _ in range(3):
            st = eng.new_state(); eng.reset_stats(); torch.cuda.synchronize(); t = time.time()
            eng.evolve(st, 0.0, 3.1); torch.cuda.synchronize(); best = min(best, time.time() - t)
        s = eng.stats()
        print(f"{name}: {3.1 / best:.1f} sim-us/s ({best * 1e3:.1f} ms), stages {s['n_applications']}, launches {s['n_launches']}, error {np.max(np.abs(st.cpu().numpy()[0] - ref)):.1e}", flush=True)
