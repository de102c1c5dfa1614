"""profiles/r04_size_batch_map.md: sim-us/s of the DEFAULT path for the full anneal on N = 10 ... 16 atoms (triangular 2 x N/2
register for even N, chain otherwise is avoided: 2 x ceil(N/2) lattice truncated to N atoms) x batch sizes.
  python tools/size_batch_map.py [N ...] > profiles/r04_size_batch_map.md"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

sizes = [int(a) for a in sys.argv[1:]] or [10, 11, 12, 13, 14, 15, 16]
batches = [1, 8, 32, 256, 512]
base = P.anneal_samples()
print("# r06: default path, full 3.1-us anneal, sim-us/s (whole batch) by register size and batch\n")
print("Register: the first N sites of a 2 x ceil(N/2) triangular lattice at the blockade radius; sequence b of a batch has its")
print("amplitude scaled by 1 - 0.3 b / (B - 1) and its detuning by the inverse factor (as bench.py).  One MI355X, best of 2.\n")
print("| N | " + " | ".join(f"B = {b}" for b in batches) + " |")
print("|---|" + "---|" * len(batches))
for n in sizes:
    coords = P.register_coords(P.triangular_rect(2, (n + 1) // 2), blockade_radius())[:n]
    cells = []
    for B in batches:
        if (2**n) * B * 16 > 6e9:
            cells.append("-"); continue
        probs = []
        for b in range(B):
            f = 1.0 - 0.3 * b / max(B - 1, 1)
            probs.append(P.make_ising_problem(coords, {"amp": base["amp"] * f, "det": base["det"] * (2.0 - f), "phase": base["phase"]}))
        with Engine.from_problems(probs, mode="sesolve") as eng:
            st = eng.new_state(); eng.evolve(st, 0.0, 0.2)
            best = None
            for rep in range(2):
                st = eng.new_state(); eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
                eng.evolve(st, 0.0, 3.1); torch.cuda.synchronize(); dt = time.time() - tic
                best = dt if best is None else min(best, dt)
            s = eng.stats()
            nrm = float(np.max(np.abs(np.linalg.norm(st.cpu().numpy(), axis=1) - 1)))
        kind = "split, one launch per run" if s["reserved"][0] > 0 and s["n_launches"] < s["n_applications"] / 20 else \
               "split passes" if s["reserved"][0] > 0 else "1 launch" if s["n_launches"] == 1 else "tiled"
        cells.append(f"{B * 3.1 / best:.1f} ({kind}; {s['n_applications']} stages, {s['n_launches']} launches; norm-1 {nrm:.0e})")
        print(f"N={n} B={B}: {cells[-1]}", file=sys.stderr, flush=True)
    print(f"| {n} | " + " | ".join(cells) + " |", flush=True)
