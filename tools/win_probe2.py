"""cProfile of the SECOND windowed Full call of a process (the first call's results still alive)."""
import os, sys, time, cProfile, pstats, io, warnings
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import blockade_radius
from test_host_logic import _inputs_from_problem
from pulser_amd import QutipEmulator, problem as P
n = int(sys.argv[1])
coords = P.register_coords(P.triangular_rect(2, (n + 1) // 2), blockade_radius())[:n]
prob = P.make_ising_problem(coords, P.anneal_samples())
keep = []
for rep in range(4):
    emu = QutipEmulator(_inputs_from_problem(prob, "ground-rydberg"), evaluation_times="Full")
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = emu.run()
    pr.disable()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: run {1e3*(t1-t0):.1f} ms + sync {1e3*(t2-t1):.1f} ms; mem allocated {torch.cuda.memory_allocated()/2**20:.0f} MB reserved {torch.cuda.memory_reserved()/2**20:.0f} MB")
    if rep == 1:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8); print(s.getvalue()[:2500])
    if rep < 2: keep.append(r)
