"""Largest density matrices (14 and 15 atoms = 4 / 16 GiB): one short dephasing mesolve,
trace, hermiticity and entries against the product of single-atom Lindblad solutions (dev probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.linalg import expm
from pulser_amd import problem as P
from pulser_amd.engine import Engine

gam = 0.5
for n in (int(a) for a in sys.argv[1:] or ["14"]):
    coords = P.register_coords(P.square_rect(1, n), 40.0)
    T = 8
    samples = {"amp": np.full(T + 1, 6.0), "det": np.full(T + 1, -2.0), "phase": np.zeros(T + 1)}
    prob = P.make_ising_problem(coords, samples, collapse_ops=[(np.sqrt(2 * gam), "sigma_rr")])
    with Engine.from_problems([prob], mode="mesolve") as eng:
        st = eng.new_state()
        t0 = time.time(); eng.evolve(st, 0.0, 0.002); torch.cuda.synchronize(); dt = time.time() - t0
        H = np.array([[2.0, 3.0], [3.0, 0.0]], dtype=complex)
        C = np.sqrt(2 * gam) * np.diag([1.0, 0.0]).astype(complex)
        I2 = np.eye(2)
        L = -1j * (np.kron(H, I2) - np.kron(I2, H.T)) + np.kron(C, C.conj()) \
            - 0.5 * np.kron(C.conj().T @ C, I2) - 0.5 * np.kron(I2, (C.conj().T @ C).T)
        r1 = (expm(L * 0.002) @ np.array([0, 0, 0, 1.0], dtype=complex)).reshape(2, 2)  # from |g><g|
        D = 1 << n
        pairs = [(0, 0), (D - 1, D - 1), (1, 2), (D - 1, 0), ((1 << (n - 1)) + 3, 5)]
        got = np.array([st[0, a, b].item() for a, b in pairs])
        ref = np.array([np.prod([r1[(a >> (n - 1 - k)) & 1, (b >> (n - 1 - k)) & 1] for k in range(n)]) for a, b in pairs])
        tr = float(torch.diagonal(st[0]).sum().real.item())
        print(f"N={n}: {dt:.2f} s, trace-1 = {tr-1:.1e}, max |rho_ab - product| = {np.max(np.abs(got-ref)):.1e}, "
              f"stats {eng.stats()}", flush=True)
