"""End-to-end wall time of QutipEmulator(...).run() for one 12-atom sequence (dev probe)."""
import sys, os, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
warnings.simplefilter("ignore")
from pulser_amd import QutipEmulator, problem as P
from pulser_amd.hamiltonian_data import single_global_channel

coords = P.register_coords(P.square_rect(1, 12), 8.692)
s = {k: v[:-1] for k, v in P.anneal_samples().items()}
inputs = single_global_channel(coords, s, P.C6_LEVEL70, extended=False)
for ev in ("Minimal", 0.1, "Full"):
    for rep in range(2):
        t0 = time.time()
        emu = QutipEmulator(inputs, evaluation_times=ev)
        t1 = time.time()
        res = emu.run()
        t2 = time.time()
        c = res.sample_final_state(1000)
        t3 = time.time()
        print(f"eval={ev} rep={rep}: ctor {t1-t0:.3f} s, run {t2-t1:.3f} s, sample {t3-t2:.3f} s, n_eval {len(emu.evaluation_times)}", flush=True)
