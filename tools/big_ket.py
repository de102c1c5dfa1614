"""Largest kets the index arithmetic allows (28 and 30 atoms = 4 / 16 GiB states): one short
evolution, norm and a few amplitudes against a product-state prediction (dev probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pulser_amd import problem as P
from pulser_amd.engine import Engine

for n in (int(a) for a in sys.argv[1:] or ["28"]):
    # atoms far apart (no interaction): every atom evolves independently -> product state
    coords = P.register_coords(P.square_rect(1, n), 40.0)
    T = 8
    t = np.arange(T + 1)
    samples = {"amp": np.full(T + 1, 6.0), "det": np.full(T + 1, -2.0), "phase": np.zeros(T + 1)}
    prob = P.make_ising_problem(coords, samples)
    with Engine.from_problems([prob], mode="sesolve") as eng:
        st = eng.new_state()
        t0 = time.time(); eng.evolve(st, 0.0, 0.004); torch.cuda.synchronize(); dt = time.time() - t0
        nrm = float(torch.linalg.vector_norm(st).item())
        # single-atom solution of the same constant drive
        H1 = np.array([[2.0, 3.0], [3.0, 0.0]])  # (r, g): -delta n_r + Omega/2 sigma_x, delta = -2, Omega = 6
        w, v = np.linalg.eigh(H1)
        u = v @ np.diag(np.exp(-1j * w * 0.004)) @ v.conj().T
        a1 = u @ np.array([0.0, 1.0])  # from |g>
        idx = [0, 1, (1 << n) - 1, (1 << (n - 1)) + 5]
        got = st[0, idx].cpu().numpy()
        ref = np.array([np.prod([a1[(i >> (n - 1 - k)) & 1] for k in range(n)]) for i in idx])
        print(f"N={n}: {dt:.2f} s, norm-1 = {nrm-1:.1e}, max |amp - product state| = {np.max(np.abs(got-ref)):.1e}, "
              f"stats {eng.stats()}", flush=True)
