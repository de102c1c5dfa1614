"""Dev probe (round 4): batches with per-atom COMPLEX drives (local addressing with phases) on k_split_reg<.., CPLX>
(method split / the default choice) against the gauged k_ket / k_traj (set_path(no_split14=True))."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import local_problem
from pulser_amd.engine import Engine

for n in (12, 13, 14):
    probs = [local_problem(n, seed=s, duration=401) for s in range(8)] * 32
    outs = {}
    for name, kw, method in (("default", {}, "auto"), ("split", {}, "split"), ("polynomial kernels", {"no_split14": True}, "auto")):
        with Engine.from_problems(probs, mode="sesolve") as eng:
            eng.set_path(False, **kw)
            st = eng.new_state(); eng.evolve(st, 0.0, 0.05, method=method)
            st = eng.new_state(); eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
            eng.evolve(st, 0.0, 0.4, method=method); torch.cuda.synchronize(); dt = time.time() - tic
            s = eng.stats(); outs[name] = st.cpu().numpy()
        print(f"{n} atoms x 256, complex drives, 400 ns, {name:18s}: {256 * 0.4 / dt:8.1f} sim-us/s, stages {s['n_applications']}, launches {s['n_launches']}, estimate {s['reserved'][0]:.1e}", flush=True)
    print(f"   max |split - polynomial| = {np.max(np.abs(outs['split'] - outs['polynomial kernels'])):.2e}, |default - polynomial| = {np.max(np.abs(outs['default'] - outs['polynomial kernels'])):.2e}")
