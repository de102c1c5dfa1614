"""Dev probe (round 4): block length of the split-operator master equation by dephasing rate: error over the FULL anneal at
12 atoms against the two-knot halves (3e-9 from the tight oracle), and ms per simulated ns at 14 atoms.  RYD_ROWS_KH."""
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
tag = f"KH={os.environ.get('RYD_ROWS_KH', '-')}"
times = np.array([0.0, 0.5, 1.3, 2.1, 3.1])
for gamma in (0.05, 0.2):
    path = f"/tmp/rows_ref2_{gamma}.npy"
    with Engine.from_problems([tri(2, 6, gamma)], mode="mesolve") as eng:
        snaps = eng.solve(eng.new_state(), times).cpu().numpy()[:, 0]
        s = eng.stats()
    if ref:
        np.save(path, snaps)
    else:
        r = np.load(path)
        print(f"{tag}: 12 atoms gamma {gamma}: max |rho - rho(KH=2)| at the 4 times = {' '.join('%.1e' % np.max(np.abs(snaps[k] - r[k])) for k in range(4))}, stages {s['n_applications']}", flush=True)
if not ref:
    with Engine.from_problems([tri(2, 7, 0.05)], mode="mesolve") as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.024)
        eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
        eng.evolve(st, 0.024, 0.120); torch.cuda.synchronize(); dt = time.time() - tic
        s = eng.stats()
        print(f"{tag}: 14 atoms, 96 ns: {dt * 1e3 / 96:.2f} ms per simulated ns ({dt * 3100 / 96:.1f} s per 3.1 us), stages {s['n_applications']}, launches {s['n_launches']}", flush=True)
