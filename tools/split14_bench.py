"""Dev probe: 14-atom batches on the one-launch split-operator kernel (k_split14_loop) against the pass-by-pass
launches (same arithmetic: identical amplitudes expected) and against k_ket (the default of the headline)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t1 = float(sys.argv[2]) if len(sys.argv) > 2 else 3.1
coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
base = P.anneal_samples()
probs = []
for b in range(B):  # different sequences: amplitude and detuning scaled per sequence (as bench.py)
    f = 1.0 - 0.3 * b / max(B - 1, 1)
    probs.append(P.make_ising_problem(coords, {"amp": base["amp"] * f, "det": base["det"] * (2.0 - f), "phase": base["phase"]}))
out = {}
for name, kw, method in (("k_ket", {}, "auto"), ("split passes", {"split_no_loop": True}, "split"), ("split one launch", {}, "split")):
    if name == "split passes" and B > 32:
        continue
    with Engine.from_problems(probs, mode="sesolve") as eng:
        eng.set_path(False, **kw)
        st = eng.new_state(); eng.evolve(st, 0.0, min(t1, 0.2), method=method)
        st = eng.new_state(); eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
        eng.evolve(st, 0.0, t1, method=method); torch.cuda.synchronize(); dt = time.time() - tic
        s = eng.stats()
        out[name] = st.cpu().numpy()
        print(f"B={B} {name:17s}: {B * t1 / dt:8.1f} sim-us/s ({dt * 1e3:.1f} ms), stages {s['n_applications']}, launches {s['n_launches']}, "
              f"steps {s['n_steps']}, estimate {s['reserved'][0]:.2e}, norm-1 {np.max(np.abs(np.linalg.norm(out[name], axis=1) - 1)):.1e}", flush=True)
for a in out:
    for b2 in out:
        if a < b2:
            print(f"max |{a} - {b2}| = {np.max(np.abs(out[a] - out[b2])):.2e}")
