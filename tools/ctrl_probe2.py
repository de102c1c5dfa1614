"""Dev probe: controller behaviour on a batch of DIFFERENT sequences (amplitude x 1 .. 0.7, detuning x 1 .. 1.3)."""
import os, sys
os.environ.setdefault("RYD_DEV", "1")  # the RYD_* A/B switches this tool reads are ignored without it (dev_common.hpp)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine
coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
base = P.anneal_samples()
for B in (2, 8):
    probs = []
    for b in range(B):
        f = 1.0 - 0.3 * b / max(B - 1, 1)
        probs.append(P.make_ising_problem(coords, {"amp": base["amp"] * f, "det": base["det"] * (2.0 - f), "phase": base["phase"]}))
    outs = {}
    for name, kw, opts in (("split", {}, {}), ("ket", {"no_split14": True, "force_ket": True}, {})):
        with Engine.from_problems(probs, mode="sesolve") as eng:
            eng.set_path(False, **kw)
            if os.environ.get('WARM'):
                st = eng.new_state(); eng.evolve(st, 0.0, 3.1); eng.reset_stats()
            st = eng.new_state(); eng.evolve(st, 0.0, 3.1, **opts); s = eng.stats()
            outs[name] = st.cpu().numpy()
        if name == "split":
            print(f"CAP={os.environ.get('RYD_SPLIT_CAP','-')} GROW={os.environ.get('RYD_SPLIT_GROW','-')} B={B}: stages {s['n_applications']}, launches {s['n_launches']}, estimate {s['reserved'][0]:.2e}, restores {s['reserved'][3]:.0f}", end="")
    print(f", max |split - k_ket| per sequence {np.max(np.abs(outs['split'] - outs['ket']), axis=1)}", flush=True)
