"""Dev probe: where the wall time of the cfg4 ensemble goes (cProfile of one run_ensemble on the GPU)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pulser_amd import NoiseModel, QutipEmulator, problem as P
from pulser_amd.distributed import run_ensemble
from pulser_amd.hamiltonian_data import single_global_channel

rb = (P.C6_LEVEL70 / (4 * 2 * np.pi / 2)) ** (1 / 6)
coords = P.register_coords(P.square_rect(1, 12), rb)
smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.005, p_false_pos=0.01, p_false_neg=0.05)

def one(seed):
    np.random.seed(seed)
    emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=1024, evaluation_times="Minimal")
    return run_ensemble(emu, dist=None, batch=256)

one(1)
t0 = time.time(); one(2); print("wall", time.time() - t0)
cProfile.run("one(3)", "/tmp/cfg4.prof")
pstats.Stats("/tmp/cfg4.prof").sort_stats("cumulative").print_stats(28)
