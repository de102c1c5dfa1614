"""NumPy probe for the next step of the master-equation path ("next"): Chin 4A blocks whose two unitary halves
are ONE sub-step each of the 6th-order split-operator composition S10 (what k_split14_loop runs) instead of K CF4 steps
with exact exponentials - error against the tight oracle over the detuning sweep (0.6 - 2.4 us) as a function of the block length.

    python tests/probes/split_rows_s10_probe.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.linalg as la
from scipy.interpolate import CubicSpline

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from helpers import blockade_radius  # noqa: E402
from ket_split_probe import scheme  # noqa: E402
from oracle import qutip_path as Q  # noqa: E402
from pulser_amd import problem as P  # noqa: E402


def run(ncol, gamma, taus=(4, 8, 16)):
    coords = P.register_coords(P.triangular_rect(2, ncol), blockade_radius())
    smp = P.anneal_samples()
    prob = P.make_ising_problem(coords, smp, collapse_ops=[(np.sqrt(2 * gamma), "sigma_rr")])
    ham = Q.build_hamiltonian(prob)
    n, dim = ham.n, 2**ham.n
    psi0 = np.zeros(dim, complex)
    psi0[-1] = 1
    # the smooth part of the sequence (the detuning sweep): kinks keep one-knot sub-steps in the library anyway
    T0, T1 = 600.0, 2400.0
    sol = Q.mesolve(ham, psi0, [0.0, T0 * 1e-3, T1 * 1e-3], **Q.TIGHT)
    rho0, ref = sol[1].reshape(dim, dim), sol[2].reshape(dim, dim)
    t_ns = np.arange(len(smp["amp"]), dtype=float)
    amp, det = CubicSpline(t_ns, smp["amp"]), CubicSpline(t_ns, smp["det"])
    det_int = det.antiderivative()
    H0 = ham.matrix(0.0).toarray()
    idx = np.arange(dim)
    occ = np.array([1 - ((idx >> (n - 1 - k)) & 1) for k in range(n)], float)
    U = P.interaction_matrix(coords, P.C6_LEVEL70)[0]
    e0 = np.einsum("is,ij,js->s", occ, np.triu(U, 1), occ)
    nexc = occ.sum(0)
    X = np.zeros((dim, dim))
    for k in range(n):
        X[idx, idx ^ (1 << k)] += 1.0
    a, b = scheme("s10_6")
    pc = np.array([[bin(x ^ y).count("1") for y in idx] for x in idx], float)

    def Dd(s):
        return np.exp(-gamma * pc * s)

    def U_s10(t0, h):  # ns
        Uacc = np.eye(dim, dtype=complex)
        tc = t0
        for i in range(len(b)):
            t1 = tc + a[i] * h
            ph = e0 * 1e-3 * (a[i] * h) - (det_int(t1) - det_int(tc)) * 1e-3 * nexc
            Uacc = np.exp(-1j * ph)[:, None] * Uacc
            tc = t1
            Uacc = la.expm(-1j * (0.5 * amp(tc) * 1e-3 * b[i] * h) * X) @ Uacc
        t1 = tc + a[-1] * h
        ph = e0 * 1e-3 * (a[-1] * h) - (det_int(t1) - det_int(tc)) * 1e-3 * nexc
        return np.exp(-1j * ph)[:, None] * Uacc

    rows = []
    for tau in taus:
        r = rho0.copy()
        k = T0
        while k < T1 - 1e-9:
            t = min(tau, T1 - k)
            U1, U2 = U_s10(k, t / 2), U_s10(k + t / 2, t / 2)
            ts = t * 1e-3
            Hmid = 0.5 * amp(k + t / 2) * X
            V = la.expm(1j * (ts**3 * gamma**2 / 72 / 2) * Hmid)
            U1, U2 = V @ U1, U2 @ V
            r = Dd(ts / 6) * r
            r = U1 @ r @ U1.conj().T
            r = Dd(2 * ts / 3) * r
            r = U2 @ r @ U2.conj().T
            r = Dd(ts / 6) * r
            k += t
        rows.append((n, gamma, tau, float(np.max(np.abs(r - ref))), 20.0 / tau))
    return rows


if __name__ == "__main__":
    print("### cfg3 anneal, Chin 4A blocks with ONE S10 sub-step per half block, vs the tight oracle: max |rho_ab - oracle|\n")
    print("| atoms | gamma | block (ns) | error | split-operator stages per ns (k_ket rows: ~9) |")
    print("|---|---|---|---|---|")
    for ncol, g in ((2, 0.05), (3, 0.05), (3, 0.5)):
        for n, gg, tau, e, spn in run(ncol, g):
            print(f"| {n} | {gg} | {tau} | {e:.1e} | {spn:.2f} |", flush=True)
