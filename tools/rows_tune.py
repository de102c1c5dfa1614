"""Dev probe (round 4): block length / composition of the split-operator master equation on k_split_reg rows:
error against the k_ket rows (<= 1e-9 from the tight oracle at 10 atoms) on an interacting 12-atom register over
0 -> 0.7 us, and ms per simulated ns at 14 atoms.  RYD_ROWS_KH (knots per half block), RYD_ROWS_S (6 / 10)."""
import os, sys, time
os.environ.setdefault("RYD_DEV", "1")  # the RYD_* A/B switches this tool reads are ignored without it (dev_common.hpp)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine


def tri(rows, cols, gamma):
    coords = P.register_coords(P.triangular_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=[(float(np.sqrt(2 * gamma)), "sigma_rr")])


ref = sys.argv[1] == "ref"
tag = f"KH={os.environ.get('RYD_ROWS_KH', '-')} S={os.environ.get('RYD_ROWS_S', '-')}"
for gamma in (0.05, 0.5):
    path = f"/tmp/rows_ref_{gamma}.npy"
    with Engine.from_problems([tri(2, 6, gamma)], mode="mesolve") as eng:
        eng.set_path(False, rows_ket=ref)
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.7)
        out = st.cpu().numpy()[0]
        s = eng.stats()
    if ref:
        np.save(path, out)
    else:
        print(f"{tag}: 12 atoms gamma {gamma}: max |split rows - k_ket rows| = {np.max(np.abs(out - np.load(path))):.2e}, stages {s['n_applications']}", flush=True)
if not ref:
    with Engine.from_problems([tri(2, 7, 0.05)], mode="mesolve") as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.012)
        eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
        eng.evolve(st, 0.012, 0.060); torch.cuda.synchronize(); dt = time.time() - tic
        s = eng.stats()
        print(f"{tag}: 14 atoms, 48 ns: {dt * 1e3 / 48:.2f} ms per simulated ns ({dt * 3100 / 48:.1f} s per 3.1 us), stages {s['n_applications']}, launches {s['n_launches']}", flush=True)
