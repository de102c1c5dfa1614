"""Ensemble density-matrix aggregation rho += sum_b |psi_b><psi_b| (dev probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import chain_problem
from pulser_amd.engine import Engine
from pulser_amd.terms import lower

for n, B in ((6, 5), (7, 33), (8, 1024), (10, 1024), (12, 256), (12, 1024), (12, 4096)):
    with Engine(lower([chain_problem(n)] * B)) as eng:
        D = 1 << n
        psi = torch.randn(B, D, dtype=torch.complex128, device=eng.device)
        acc = torch.zeros(D, D, dtype=torch.complex128, device=eng.device)
        w = np.random.default_rng(1).uniform(0.5, 1.5, B)
        eng.outer_accumulate(psi, acc, w); torch.cuda.synchronize()
        wt = torch.from_numpy(w).to(eng.device)
        refw = (psi * wt[:, None]).T @ psi.conj()
        errw = float((acc - refw).abs().max() / refw.abs().max())
        acc.zero_()
        t0 = time.time(); eng.outer_accumulate(psi, acc); torch.cuda.synchronize(); dt = time.time() - t0
        t0 = time.time(); ref = psi.T @ psi.conj(); torch.cuda.synchronize(); dt2 = time.time() - t0
        t0 = time.time(); ref = psi.T @ psi.conj(); torch.cuda.synchronize(); dt2 = time.time() - t0
        err = max(errw, float((acc - ref).abs().max() / ref.abs().max()))
        fl = 8.0 * D * D * B
        print(f"N={n} B={B}: outer_accumulate {dt*1e3:.2f} ms ({fl/dt/1e12:.2f} TFLOP/s), rocBLAS zgemm {dt2*1e3:.2f} ms "
              f"({fl/dt2/1e12:.2f} TFLOP/s), rel err {err:.1e}", flush=True)
