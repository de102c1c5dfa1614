"""cfg4 end to end on one GPU: 1024 noisy trajectories of the 12-atom anneal
through the QutipEmulator front-end (dev timing probe)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import load_fixture
from test_host_logic import _chain12_inputs
from pulser_amd import NoiseModel, QutipEmulator
warnings.simplefilter("ignore")
_, extra = load_fixture("cfg4_chain12_noise.npz")
nm = NoiseModel(samples_per_run=1, **extra["noise_model"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for rep in range(2):
    np.random.seed(0)
    t0 = time.time()
    emu = QutipEmulator(_chain12_inputs(extra), noise_model=nm, n_trajectories=n, evaluation_times="Minimal")
    t1 = time.time()
    res = emu.run()
    torch.cuda.synchronize()
    t2 = time.time()
    print(f"run {rep}: ctor {t1-t0:.3f}s run {t2-t1:.3f}s -> {n*3.1/(t2-t1):.1f} sim-us/s end-to-end; top counts {res[-1].bitstring_counts.most_common(3) if hasattr(res[-1].bitstring_counts,'most_common') else ''}", flush=True)
if os.environ.get("RYD_PROFILE"):
    import cProfile, pstats
    np.random.seed(0)
    emu = QutipEmulator(_chain12_inputs(extra), noise_model=nm, n_trajectories=n, evaluation_times="Minimal")
    pr = cProfile.Profile(); pr.enable(); emu.run(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
