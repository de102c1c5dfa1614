import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from helpers import fuzz_case
from pulser_amd.engine import Engine
probs, desc = fuzz_case(2685)
with Engine.from_problems(probs, mode="sesolve") as eng:
    # (a) norms of the error vector along the path
    for t in (96, 112, 120, 128, 136, 144, 160, 183):
        ref = eng.new_state(); eng.evolve(ref, 0.0, t * 1e-3, method="taylor", tol=1e-13, magnus_tol=1e-13)
        st = eng.new_state(); eng.evolve(st, 0.0, t * 1e-3)
        d = st - ref
        print(f"t {t}: max {float(d.abs().max()):.2e}  2-norm {float(torch.linalg.vector_norm(d)):.2e}")
    # (b) the stretch 120 -> 144 alone, from the reference state at 120
    r0 = eng.new_state(); eng.evolve(r0, 0.0, 0.120, method="taylor", tol=1e-13, magnus_tol=1e-13)
    for (a, b) in ((120, 128), (128, 136), (136, 144), (120, 144), (96, 120)):
        ra = eng.new_state(); eng.evolve(ra, 0.0, a * 1e-3, method="taylor", tol=1e-13, magnus_tol=1e-13)
        rb = ra.clone(); eng.evolve(rb, a * 1e-3, b * 1e-3, method="taylor", tol=1e-13, magnus_tol=1e-13)
        sb = ra.clone(); eng.reset_stats(); eng.evolve(sb, a * 1e-3, b * 1e-3); s = eng.stats()
        d = sb - rb
        print(f"stretch {a} -> {b} from the reference state: max {float(d.abs().max()):.2e} 2-norm {float(torch.linalg.vector_norm(d)):.2e} est {s['reserved'][0]:.2e} stages {s['n_applications']}")
