"""Dev probe: split-operator passes, S6 with one-knot sub-steps (round 2) against S10 with multi-knot sub-steps:
accuracy against CF4 + Taylor (tol 1e-13) and wall time, 14 / 16 / 20 atoms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

SHAPES = {14: ("tri", 2, 7), 16: ("rect", 4, 4), 20: ("rect", 4, 5), 22: ("rect", 2, 11)}
for n in [int(v) for v in sys.argv[1:]] or [14, 16, 20]:
    kind, r, c = SHAPES[n]
    pat = P.triangular_rect(r, c) if kind == "tri" else P.square_rect(r, c)
    prob = P.make_ising_problem(P.register_coords(pat, blockade_radius()), P.anneal_samples())
    t0, t1 = (0.0, 3.1) if n <= 16 else (0.4, 0.7)
    if os.environ.get("SLICE"):
        t0, t1 = [float(v) for v in os.environ["SLICE"].split(",")]
    with Engine.from_problems([prob], mode="sesolve") as eng:
        eng.set_path(True, no_ket=True, no_split=True)
        start = eng.new_state()
        if t0 > 0:
            eng.evolve(start, 0.0, t0, method="taylor", tol=1e-12)
        ref = start.clone()
        eng.evolve(ref, t0, t1, method="taylor", tol=1e-13, magnus_tol=1e-12)
        ref = ref.cpu().numpy()
    for s6 in (True, False):
        with Engine.from_problems([prob], mode="sesolve") as eng:
            eng.set_path(False, split_s6=s6)
            st = start.clone()
            if not os.environ.get("NOWARM"):
                eng.evolve(st, t0, min(t0 + 0.05, t1), method="split")  # warm-up (and the controller's first checks)
            st = start.clone(); eng.reset_stats(); torch.cuda.synchronize(); tic = time.time()
            eng.evolve(st, t0, t1, method="split"); torch.cuda.synchronize(); dt = time.time() - tic
            s = eng.stats()
            err = np.max(np.abs(st.cpu().numpy() - ref))
            print(f"N={n} {'S6 one-knot' if s6 else 'S10 multi-knot'}: {(t1 - t0) / dt:.2f} sim-us/s ({dt * 1e3:.1f} ms), stages {s['n_applications']}, "
                  f"launches {s['n_launches']}, steps {s['n_steps']}, |psi - taylor| {err:.2e}, estimate {s['reserved'][0]:.2e}, tau {s['reserved'][2] * 1e3:.2f} ns, restores {s['reserved'][3]:.0f}", flush=True)
