"""1024 trajectories with high-frequency detuning noise (20 frequencies): host lowering + GPU solve (dev probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pulser_amd import problem as P
from pulser_amd.engine import Engine
from pulser_amd.hamiltonian_data import single_global_channel, HamiltonianData
from pulser_amd.noise_model import NoiseModel

coords = P.register_coords(P.square_rect(1, 12), 8.692)
s = {k: v[:-1] for k, v in P.anneal_samples().items()}
inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
for nf in (0, 20):
    kw = dict(temperature=50.0, amp_sigma=0.05, detuning_sigma=0.2)
    if nf:
        kw.update(detuning_hf_psd=tuple(np.linspace(2, 0.1, nf + 1)), detuning_hf_omegas=tuple(np.linspace(5, 200, nf + 1)))
    np.random.seed(0)
    hd = HamiltonianData(inputs.extend_duration(3101), NoiseModel(**kw), 1024)
    t0 = time.time()
    tabs = hd.device_tables(hd.noise_trajectories[:256], 1.0)
    t1 = time.time()
    with Engine(tabs, mode="sesolve") as eng:
        st = eng.new_state(); eng.evolve(st, 0.0, 0.01); torch.cuda.synchronize()
        eng.reset_stats(); st = eng.new_state(); t2 = time.time(); eng.evolve(st, 0.0, 3.1); torch.cuda.synchronize(); t3 = time.time(); stats = eng.stats()
    print(f"hf frequencies {nf}: stats {eng.stats() if False else stats}; lowering 256 trajectories {t1-t0:.3f} s, GPU solve {t3-t2:.3f} s "
          f"({256*3.1/(t3-t2):.0f} sim-us/s)", flush=True)
