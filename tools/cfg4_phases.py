"""Where the wall time of one 512-trajectory block solve goes (cfg4): Engine construction, new_state, solve, occupations."""
import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from bench import blockade_radius
from pulser_amd import NoiseModel, QutipEmulator, problem as P
from pulser_amd.hamiltonian_data import single_global_channel
from pulser_amd.engine import Engine
coords = P.register_coords(P.square_rect(1, 12), blockade_radius())
smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
inputs = single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)
nm = NoiseModel(temperature=50.0, amp_sigma=0.05, state_prep_error=0.005, p_false_pos=0.01, p_false_neg=0.05)
np.random.seed(0)
emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=1024, evaluation_times="Minimal")
hd = emu._hamiltonian_data
trajs = hd.noise_trajectories
tables = hd.device_tables(trajs[:512], emu._sampling_rate)
times = emu._eval_times_array
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng = Engine(tables, mode="sesolve")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    state = eng.new_state(np.asarray(emu._initial_state).reshape(1, -1))
    first = state.clone()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    snaps = eng.solve(state, times, store=True)
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    occ = torch.stack([eng.occupations(first)] + [eng.occupations(snaps[i]) for i in range(len(times) - 1)]).cpu().numpy()
    t5 = time.perf_counter()
    st = eng.stats()
    eng.close()
    t6 = time.perf_counter()
    print(f"engine {1e3*(t1-t0):.2f} ms, state {1e3*(t2-t1):.2f}, solve {1e3*(t3-t2):.2f} (+sync {1e3*(t4-t3):.2f}), occ {1e3*(t5-t4):.2f}, close {1e3*(t6-t5):.2f}; "
          f"stages {st['n_applications']} launches {st['n_launches']}")
