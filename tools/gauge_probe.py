"""Dev probe: the 13-atom noisy sequence with a pulse phase (tests/test_gpu_ket.py) on every path against a tight CF4 + Taylor run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pulser_amd import problem as P, NoiseModel, QutipEmulator
from pulser_amd.hamiltonian_data import single_global_channel
from pulser_amd.engine import Engine

n = 13
coords = P.register_coords(P.square_rect(1, n), 8.0)
s = {k: np.asarray(v)[:300].copy() for k, v in P.anneal_samples().items()}
s["phase"] = np.full(300, float(os.environ.get("PROBE_PHASE", "0.7")))
inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
nm = NoiseModel(temperature=50.0, amp_sigma=0.05)
np.random.seed(4)
emu = QutipEmulator(inputs, noise_model=nm, n_trajectories=16, evaluation_times="Minimal")
hd = emu._hamiltonian_data
tables = hd.device_tables(hd.noise_trajectories, emu._sampling_rate)
times = np.asarray(emu._eval_times_array)
print("times", times)
outs = {}
for name, kw, ev in (("tight", dict(no_ket=True), dict(method="taylor", tol=1e-13, magnus_tol=1e-12)), ("k_ket gauge", dict(no_split14=True), {}),
                     ("no_ket", dict(no_ket=True), {}), ("default", {}, {}), ("default s6", dict(split_s6=True), {}), ("default fixed", dict(split_fixed=True), {})):
    with Engine(tables, mode="sesolve") as eng:
        eng.set_path(False, **kw)
        st = eng.new_state()
        for a, b in zip(times[:-1], times[1:]):
            eng.evolve(st, float(a), float(b), **ev)
        outs[name] = st.cpu().numpy()
        stt = eng.stats()
        print(f"reserved {[float('%.3g' % v) for v in stt['reserved'][:4]]}", end=" ")
        print(f"{name}: launches {stt['n_launches']} stages {stt['n_applications']} steps {stt['n_steps']}; max |psi - tight| = {np.max(np.abs(outs[name] - outs['tight'])):.2e}", flush=True)
