"""Dev probe (round 4): the split-operator master equation with its row passes on k_split_reg (default from 12 atoms)
against the same on k_ket (set_path(rows_ket=True)): agreement on an interacting 12-atom register (dephasing 0.05 and
0.5 / us: the commutator kick of the 4th-order splitting matters at the second rate) and wall time of a cfg3 slice.
  python tools/rows_bench.py [slice_ns]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def tri(rows, cols, gamma):
    coords = P.register_coords(P.triangular_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=[(float(np.sqrt(2 * gamma)), "sigma_rr")])


for gamma in (0.05, 0.5):
    outs = {}
    for name, kw in (("k_split_reg", {}), ("k_ket", {"rows_ket": True})):
        with Engine.from_problems([tri(2, 6, gamma)], mode="mesolve") as eng:
            eng.set_path(False, **kw)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.7)
            outs[name] = st.cpu().numpy()[0]
            s = eng.stats()
        print(f"12 atoms, gamma {gamma}, 0 -> 0.7 us on {name}: trace-1 {np.trace(outs[name]).real - 1:.1e}, "
              f"stages {s['n_applications']}, launches {s['n_launches']}", flush=True)
    print(f"   max |k_split_reg - k_ket| = {np.max(np.abs(outs['k_split_reg'] - outs['k_ket'])):.2e}", flush=True)

for name, kw in (("k_split_reg", {}), ("k_ket", {"rows_ket": True})):
    with Engine.from_problems([tri(2, 7, 0.05)], mode="mesolve") as eng:
        eng.set_path(False, **kw)
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.004)
        eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
        eng.evolve(st, 0.004, 0.004 + ns * 1e-3); torch.cuda.synchronize(); dt = time.time() - tic
        s = eng.stats()
        tr = float(torch.diagonal(st[0]).real.sum().item())
        print(f"14 atoms, {ns} ns on {name}: {dt * 1e3 / ns:.2f} ms per simulated ns ({dt * 3100 / ns:.1f} s per 3.1 us), "
              f"stages {s['n_applications']}, launches {s['n_launches']}, trace-1 {tr - 1:.1e}", flush=True)
