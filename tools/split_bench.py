"""Per-pass time of k_split (fixed sub-step, no controller) and throughput with the controller.
python tools/split_bench.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

SHAPES = {14: (2, 7), 15: (3, 5), 16: (4, 4), 17: (1, 17), 18: (3, 6), 19: (1, 19), 20: (4, 5), 21: (3, 7), 22: (2, 11), 23: (1, 23), 24: (4, 6), 25: (5, 5)}

def rect(rows, cols):
    coords = P.register_coords(P.square_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples())

for n in [int(v) for v in sys.argv[1:] if v.isdigit()] or ([] if "batch" in sys.argv else [16, 20, 22]):
    with Engine.from_problems([rect(*SHAPES[n])], mode="sesolve") as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.01, method="split")
        ns = 200 if n <= 20 else 50
        for fixed in (True, False):
            eng.set_path(False, split_fixed=fixed, split_small_tiles=bool(os.environ.get("RYD_BENCH_SMALL")))
            eng.reset_stats()
            torch.cuda.synchronize(); t0 = time.time()
            eng.evolve(st, 1.0, 1.0 + ns * 1e-3, method="split")
            torch.cuda.synchronize(); dt = time.time() - t0
            s = eng.stats()
            line = f"N={n} fixed={fixed}: {ns} ns in {dt*1e3:.1f} ms = {dt/ns*1e6:.1f} us/ns ({ns*1e-3/dt:.2f} sim-us/s); launches {s['n_launches']} -> {dt/s['n_launches']*1e6:.1f} us wall per pass"
            if fixed:
                eng.set_kernel_timing(True)
                eng.evolve(st, 1.0, 1.0 + 20e-3, method="split")
                torch.cuda.synchronize()
                ms, nl = eng.kernel_timing()
                eng.set_kernel_timing(False)
                line += f"; kernel {ms/nl*1e3:.1f} us per pass (HIP events, {nl} launches)"
            else:
                line += f"; est_err {s['reserved'][0]:.2e} tau {s['reserved'][2]:.2e}"
            print(line, flush=True)


def batch_cases():
    """Batches: 256 x 12 atoms (one-launch loop vs the LDS-resident k_traj) and 256 x 14 atoms
    (streaming passes vs the register-resident k_ket)."""
    chain = P.make_ising_problem(P.register_coords(P.square_rect(1, 12), blockade_radius()), P.anneal_samples())
    tri = P.make_ising_problem(P.register_coords(P.triangular_rect(2, 7), blockade_radius()), P.anneal_samples())
    for label, prob, t0, t1 in (("256 x 12 atoms, full 3.1 us", chain, 0.0, 3.1), ("256 x 14 atoms, 100 ns at 1 us", tri, 1.0, 1.1)):
        for method in ("auto", "split"):
            with Engine.from_problems([prob] * 256, mode="sesolve") as eng:
                if method == "split":
                    eng.set_path(False, no_ket=True)
                st = eng.new_state()
                eng.evolve(st, t0, t0 + 0.01, method=method)
                st = eng.new_state()
                eng.reset_stats()
                torch.cuda.synchronize(); t_0 = time.time()
                eng.evolve(st, t0, t1, method=method)
                torch.cuda.synchronize(); dt = time.time() - t_0
                s = eng.stats()
                print(f"{label} method={method}: {dt*1e3:.1f} ms -> {256*(t1-t0)/dt:.0f} sim-us/s; launches {s['n_launches']} "
                      f"stages {s['n_applications']} est_err {s['reserved'][0]:.2e}", flush=True)


if "batch" in sys.argv:
    batch_cases()
