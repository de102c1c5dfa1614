"""Dev probe (round 4): microseconds per stage of the one-launch 14-atom kernels at a FIXED stage count (no step-size
control: 6 stages per knot interval), so that knock-out builds (wrong amplitudes) can be timed too.
  [RYD_LIB=build/variants/x.so] python tools/lane_time.py [B] [turns]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
turns = len(sys.argv) > 2 and sys.argv[2] == "turns"
coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
base = P.anneal_samples()
probs = []
for b in range(B):
    f = 1.0 - 0.3 * b / max(B - 1, 1)
    probs.append(P.make_ising_problem(coords, {"amp": base["amp"] * f, "det": base["det"] * (2.0 - f), "phase": base["phase"]}))
with Engine.from_problems(probs, mode="sesolve") as eng:
    eng.set_path(False, split_fixed=True, split_turns=turns, force_ket=(B < 8))
    st = eng.new_state(); eng.evolve(st, 0.0, 0.6, method="split")
    best = None
    for rep in range(3):
        s0 = st.clone(); eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
        eng.evolve(s0, 0.6, 1.6, method="split"); torch.cuda.synchronize(); dt = time.time() - tic
        best = dt if best is None else min(best, dt)
    s = eng.stats()
    print(f"{os.environ.get('RYD_LIB', 'default'):28s} {'turns' if turns else 'lane ':5s} B={B}: {best * 1e6 / s['n_applications']:.3f} us per stage "
          f"({s['n_applications']} stages, {s['n_launches']} launches, {best * 1e3:.1f} ms)", flush=True)
