"""Dev probe (round 4): 14-atom batches on k_split_reg (lane bits over the DPP crossbar / permlane swaps, one LDS
pass per stage) against k_split14_loop (round 3: two LDS turns per stage) and the pass-by-pass launches.
  python tools/lane_bench.py [B] [t1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t1 = float(sys.argv[2]) if len(sys.argv) > 2 else 3.1
coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
base = P.anneal_samples()


def problems(n):
    out = []
    for b in range(n):
        f = 1.0 - 0.3 * b / max(n - 1, 1)
        out.append(P.make_ising_problem(coords, {"amp": base["amp"] * f, "det": base["det"] * (2.0 - f), "phase": base["phase"]}))
    return out


# 1. same stages, same order: rounding-level agreement with the pass-by-pass launches (8 sequences, two slices)
probs = problems(8)
for t0, te in ((0.0, 0.62), (2.4, 3.1)):
    outs = {}
    for name, kw in (("lane", {}), ("turns", {"split_turns": True}), ("passes", {"split_no_loop": True})):
        with Engine.from_problems(probs, mode="sesolve") as eng:
            eng.set_path(False, **kw)
            st = eng.new_state()
            if t0 > 0:
                eng.evolve(st, 0.0, t0, method="taylor")
            eng.evolve(st, t0, te, method="split")
            outs[name] = st.cpu().numpy()
    print(f"[{t0}, {te}] us: max |lane - passes| = {np.max(np.abs(outs['lane'] - outs['passes'])):.2e}, "
          f"|turns - passes| = {np.max(np.abs(outs['turns'] - outs['passes'])):.2e}, "
          f"norm-1 (lane) = {np.max(np.abs(np.linalg.norm(outs['lane'], axis=1) - 1)):.1e}", flush=True)

# 2. the headline batch
probs = problems(B)
res = {}
for name, kw in (("k_split_reg", {}), ("k_split14_loop", {"split_turns": True})):
    with Engine.from_problems(probs, mode="sesolve") as eng:
        eng.set_path(False, **kw)
        st = eng.new_state(); eng.evolve(st, 0.0, min(t1, 0.2))
        best = None
        for rep in range(3):
            st = eng.new_state(); eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
            eng.evolve(st, 0.0, t1); torch.cuda.synchronize(); dt = time.time() - tic
            best = dt if best is None else min(best, dt)
        s = eng.stats()
        res[name] = st.cpu().numpy()
        print(f"B={B} {name:15s}: {B * t1 / best:8.1f} sim-us/s ({best * 1e3:.1f} ms), stages {s['n_applications']}, launches {s['n_launches']}, "
              f"{best * 1e6 / max(s['n_applications'], 1):.2f} us per stage (wall), estimate {s['reserved'][0]:.2e}, "
              f"norm-1 {np.max(np.abs(np.linalg.norm(res[name], axis=1) - 1)):.1e}", flush=True)
print(f"max |lane - turns| = {np.max(np.abs(res['k_split_reg'] - res['k_split14_loop'])):.2e}")
