#!/usr/bin/env python
"""cProfile of one cfg4 ensemble (1024 noisy 12-atom trajectories through run_ensemble): where the host time goes.
CFG4_BATCH = trajectories per engine batch (default 512)."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import blockade_radius  # noqa: E402
from pulser_amd import NoiseModel, QutipEmulator, problem as P  # noqa: E402
from pulser_amd.distributed import run_ensemble  # noqa: E402
from pulser_amd.hamiltonian_data import single_global_channel  # noqa: E402

coords = P.register_coords(P.square_rect(1, 12), blockade_radius())
smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.005, p_false_pos=0.01, p_false_neg=0.05)


def one(seed):
    np.random.seed(seed)
    emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=1024, evaluation_times="Minimal")
    return run_ensemble(emu, dist=None, batch=int(os.environ.get("CFG4_BATCH", "512")))


one(100)
torch.cuda.synchronize()
for rep in range(6):
    tic = time.perf_counter()
    r = one(rep)
    torch.cuda.synchronize()
    print("ensemble", (time.perf_counter() - tic) * 1e3, "ms", {k: round(v, 2) for k, v in r["timings"].items()})
pr = cProfile.Profile()
pr.enable()
one(5)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
