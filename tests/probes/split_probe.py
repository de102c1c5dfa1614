"""NumPy probe of the operator splitting behind the master-equation row path (host_ket.hpp):
error of Strang and of Chin's 4th-order scheme 4A (with / without the exact commutator kick)
against the tight oracle / the exact single-atom solution, as a function of the block length.

    python tests/probes/split_probe.py            # writes the tables of profiles/r02_split_probe.md to stdout
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.linalg as la

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import blockade_radius  # noqa: E402
from oracle import qutip_path as Q  # noqa: E402
from pulser_amd import problem as P  # noqa: E402

S3 = np.sqrt(3.0)
C1, C2 = 0.5 - S3 / 6, 0.5 + S3 / 6
A1, A2 = 0.25 + S3 / 6, 0.25 - S3 / 6


def single_atom(gamma, omega, delta, tau_ns, t_ns=3100):
    """Constant resonant-ish drive on one atom: the worst case for the splitting error."""
    H = np.array([[-delta, omega / 2], [omega / 2, 0.0]], complex)
    Hd = np.array([[0, omega / 2], [omega / 2, 0.0]], complex)
    C = np.sqrt(2 * gamma) * np.diag([1.0, 0.0]).astype(complex)
    I2 = np.eye(2)
    L = (-1j * (np.kron(H, I2) - np.kron(I2, H.T)) + np.kron(C, C.conj())
         - 0.5 * np.kron(C.conj().T @ C, I2) - 0.5 * np.kron(I2, (C.conj().T @ C).T))
    tau = tau_ns * 1e-3
    nblk = int(t_ns / tau_ns)
    exact = (la.expm(L * tau * nblk) @ np.array([0, 0, 0, 1.0], complex)).reshape(2, 2)
    Uh = la.expm(-1j * H * tau / 2)
    V = la.expm(1j * (tau**3 * gamma**2 / 72 / 2) * Hd)

    def D(s):
        e = np.exp(-gamma * s)
        return np.array([[1, e], [e, 1]])

    out = []
    for name in ("strang", "chin4a_no_kick", "chin4a"):
        r = np.array([[0, 0], [0, 1.0]], complex)
        for _ in range(nblk):
            if name == "strang":
                U = Uh @ Uh
                r = D(tau / 2) * (U @ (D(tau / 2) * r) @ U.conj().T)
            else:
                W1 = V @ Uh if name == "chin4a" else Uh
                W2 = Uh @ V if name == "chin4a" else Uh
                r = D(tau / 6) * r
                r = W1 @ r @ W1.conj().T
                r = D(2 * tau / 3) * r
                r = W2 @ r @ W2.conj().T
                r = D(tau / 6) * r
        out.append(np.abs(r - exact).max())
    return out


def anneal(ncol, gamma, taus=(2, 4, 8, 16)):
    """2 x ncol triangular register, the cfg3 anneal, against the tight oracle (zvode rtol 1e-13)."""
    coords = P.register_coords(P.triangular_rect(2, ncol), blockade_radius())
    prob = P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=[(np.sqrt(2 * gamma), "sigma_rr")])
    ham = Q.build_hamiltonian(prob)
    n, dim = ham.n, 2**ham.n
    psi0 = np.zeros(dim, complex)
    psi0[-1] = 1

    def Hm(t):
        return ham.matrix(t).toarray()

    def U_cf4(t, h):
        H1, H2 = Hm(t + C1 * h), Hm(t + C2 * h)
        return la.expm(-1j * h * (A2 * H1 + A1 * H2)) @ la.expm(-1j * h * (A1 * H1 + A2 * H2))

    idx = np.arange(dim)
    pc = np.array([[bin(a ^ b).count("1") for b in idx] for a in idx], float)

    def D(s):
        return np.exp(-gamma * pc * s)

    ref = Q.mesolve(ham, psi0, [0.0, 3.1], **Q.TIGHT)[-1]
    Us = [U_cf4(k * 1e-3, 1e-3) for k in range(3100)]
    rows = []
    for tau in taus:
        kh = tau // 2
        errs = []
        for kick in (False, True):
            r = np.outer(psi0, psi0.conj())
            k = 0
            while k < 3100:
                k1 = min(kh, 3100 - k)
                k2 = min(kh, 3100 - k - k1)
                U1 = np.eye(dim)
                for j in range(k1):
                    U1 = Us[k + j] @ U1
                if k2 == 0:
                    t1 = k1 * 1e-3
                    r = D(t1 / 2) * (U1 @ (D(t1 / 2) * r) @ U1.conj().T)
                    k += k1
                    continue
                U2 = np.eye(dim)
                for j in range(k2):
                    U2 = Us[k + k1 + j] @ U2
                t = (k1 + k2) * 1e-3
                if kick:
                    Hmid = Hm((k + k1) * 1e-3)
                    V = la.expm(1j * (t**3 * gamma**2 / 72 / 2) * (Hmid - np.diag(np.diag(Hmid))))
                    U1, U2 = V @ U1, U2 @ V
                r = D(t / 6) * r
                r = U1 @ r @ U1.conj().T
                r = D(2 * t / 3) * r
                r = U2 @ r @ U2.conj().T
                r = D(t / 6) * r
                k += k1 + k2
            errs.append(np.max(np.abs(r - ref)))
        rows.append((n, gamma, tau, errs[0], errs[1]))
    return rows


if __name__ == "__main__":
    print("### single atom, constant drive, delta = -2, 3.1 us: max |rho - exact|\n")
    print("| gamma (1/us) | Omega (rad/us) | block tau (ns) | Strang | Chin 4A without kick | Chin 4A with the commutator kick |")
    print("|---|---|---|---|---|---|")
    for g in (0.05, 0.5):
        for om in (6.0, 25.0):
            for tau in (2, 4, 8, 16):
                e = single_atom(g, om, -2.0, tau)
                print(f"| {g} | {om} | {tau} | {e[0]:.1e} | {e[1]:.1e} | {e[2]:.1e} |", flush=True)
    print("\n### cfg3 anneal on 2 x n triangular registers (interacting), 3.1 us, vs the tight oracle: max |rho_ab - oracle|\n")
    print("| atoms | gamma | block tau (ns) | Chin 4A without kick | with kick |")
    print("|---|---|---|---|---|")
    for ncol, g in ((2, 0.05), (2, 0.5), (3, 0.05)):
        for n, gg, tau, e0, e1 in anneal(ncol, g):
            print(f"| {n} | {gg} | {tau} | {e0:.1e} | {e1:.1e} |", flush=True)
