"""Dev probe (round 4): the one-launch 14-atom kernels against the pass-by-pass launches (same stages, same order:
rounding-level agreement expected).  RYD_SPLIT_NR selects the variant (5 default, 6)."""
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
probs = []
for b in range(8):
    f = 1.0 - 0.3 * b / 7
    probs.append(P.make_ising_problem(coords, {"amp": base["amp"] * f, "det": base["det"] * (2.0 - f), "phase": base["phase"]}))
for t0, te in ((0.0, 0.62), (2.4, 3.1)):
    outs = {}
    for name, kw in (("one launch", {}), ("passes", {"split_no_loop": True})):
        with Engine.from_problems(probs, mode="sesolve") as eng:
            eng.set_path(False, **kw)
            st = eng.new_state()
            if t0 > 0:
                eng.evolve(st, 0.0, t0, method="taylor")
            eng.evolve(st, t0, te, method="split")
            outs[name] = st.cpu().numpy()
    print(f"NR={os.environ.get('RYD_SPLIT_NR', 'default')} [{t0}, {te}] us: max |one launch - passes| = "
          f"{np.max(np.abs(outs['one launch'] - outs['passes'])):.2e}, norm-1 = "
          f"{np.max(np.abs(np.linalg.norm(outs['one launch'], axis=1) - 1)):.1e}", flush=True)
