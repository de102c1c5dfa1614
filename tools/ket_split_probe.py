"""NumPy model of a split-operator ket propagator  exp(-i tau (D + X)) ~ prod_i D(a_i tau) R(b_i tau)
for the Ising problem (D = interaction + detuning diagonal, X = sum_j drive_j: a product of single-atom
rotations), against a tight CF4 + Taylor reference.  Chooses scheme / step for k_split without a GPU.

    python tools/ket_split_probe.py ROWS COLS [tri|rect] [tau_ns ...]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
from scipy.interpolate import CubicSpline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pulser_amd import problem as P  # noqa: E402

S3 = np.sqrt(3.0)
C1, C2 = 0.5 - S3 / 6, 0.5 + S3 / 6
A1, A2 = 0.25 + S3 / 6, 0.25 - S3 / 6


def blockade_radius():
    return (P.C6_LEVEL70 / (4 * 2 * np.pi / 2)) ** (1 / 6)


# Symmetric compositions  A(a1) B(b1) A(a2) ... B(b1) A(a1)  (Blanes & Moan 2002, tables 2-3), A = D here
def scheme(name):
    if name == "strang":
        return np.array([0.5, 0.5]), np.array([1.0])
    if name == "s6_4":
        a = [0.0792036964311957, 0.353172906049774, -0.0420650803577195]
        b = [0.209515106613362, -0.143851773179818]
        a4 = 1 - 2 * sum(a)
        b3 = 0.5 - sum(b)
        return np.array(a + [a4] + a[::-1]), np.array(b + [b3, b3] + b[::-1])
    if name == "s10_6":
        a = [0.0502627644003922, 0.413514300428344, 0.0450798897943977, -0.188054853819569, 0.541960678450780]
        b = [0.148816447901042, -0.132385865767784, 0.067307604692185, 0.432666402578175]
        a6 = 1 - 2 * sum(a)
        b5 = 0.5 - sum(b)
        return np.array(a + [a6] + a[::-1]), np.array(b + [b5, b5] + b[::-1])
    raise ValueError(name)


class Prob:
    def __init__(self, rows, cols, kind):
        pat = P.triangular_rect(rows, cols) if kind == "tri" else P.square_rect(rows, cols)
        coords = P.register_coords(pat, blockade_radius())
        s = P.anneal_samples()
        self.n = n = len(coords)
        U = P.interaction_matrix(coords, P.C6_LEVEL70)[0]
        idx = np.arange(1 << n)
        # qubit k <-> bit n-1-k; state 'r' = bit 0 (eigenbasis ["r", "g"]): n_k = 1 - bit
        occ = np.array([1 - ((idx >> (n - 1 - k)) & 1) for k in range(n)], dtype=float)
        self.occ = occ
        self.e0 = np.einsum("is,ij,js->s", occ, np.triu(U, 1), occ)
        self.nexc = occ.sum(0)
        T = len(s["amp"])
        t = np.arange(T, dtype=float)
        self.T = T - 1
        self.amp = CubicSpline(t, s["amp"])
        self.det = CubicSpline(t, s["det"])
        self.det_int = self.det.antiderivative()

    def diag(self, t):
        return self.e0 - self.det(t) * 1e-3 * self.nexc  # rad / ns (samples in rad/us)

    def apply_h(self, psi, t):
        n = self.n
        out = (self.e0 * 1e-3 - self.det(t) * 1e-3 * self.nexc) * psi
        w = 0.5 * self.amp(t) * 1e-3
        x = psi.reshape((2,) * n)
        acc = np.zeros_like(x)
        for k in range(n):
            acc += np.flip(x, axis=k)
        return out + w * acc.reshape(-1)


def reference(pr, t_end, h=0.25, order=18):
    n = pr.n
    psi = np.zeros(1 << n, dtype=complex)
    psi[-1] = 1.0
    t = 0.0
    snaps = {}
    while t < t_end - 1e-9:
        for (u1, u2) in ((A1, A2), (A2, A1)):
            term = psi.copy()
            acc = psi.copy()
            for j in range(1, order + 1):
                term = (-1j * h / j) * (u1 * pr.apply_h(term, t + C1 * h) + u2 * pr.apply_h(term, t + C2 * h))
                acc += term
            psi = acc
        t += h
        if abs(t - round(t)) < 1e-9 and int(round(t)) % 100 == 0:
            snaps[int(round(t))] = psi.copy()
    return psi, snaps


def split_run(pr, t_end, tau, name, snaps_at=()):
    n = pr.n
    a, b = scheme(name)
    psi = np.zeros(1 << n, dtype=complex)
    psi[-1] = 1.0
    t = 0.0
    x = None
    out = {}
    nsteps = int(round(t_end / tau))
    for s in range(nsteps):
        tc = s * tau
        for i in range(len(b)):
            # D over [tc, tc + a_i tau]: exact phase (time advances with D)
            t1 = tc + a[i] * tau
            ph = pr.e0 * 1e-3 * (a[i] * tau) - (pr.det_int(t1) - pr.det_int(tc)) * 1e-3 * pr.nexc
            psi = psi * np.exp(-1j * ph)
            tc = t1
            # R: frozen drive at tc
            th = 0.5 * pr.amp(tc) * 1e-3 * b[i] * tau
            c, sn = np.cos(th), np.sin(th)
            x = psi.reshape((2,) * n)
            for k in range(n):
                x = c * x - 1j * sn * np.flip(x, axis=k)
            psi = x.reshape(-1)
        t1 = tc + a[-1] * tau
        ph = pr.e0 * 1e-3 * (a[-1] * tau) - (pr.det_int(t1) - pr.det_int(tc)) * 1e-3 * pr.nexc
        psi = psi * np.exp(-1j * ph)
        tt = (s + 1) * tau
        if abs(tt - round(tt)) < 1e-9 and int(round(tt)) in snaps_at:
            out[int(round(tt))] = psi.copy()
    return psi, out


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    kind = sys.argv[3] if len(sys.argv) > 3 else "rect"
    taus = [float(v) for v in sys.argv[4:]] or [1.0, 0.5]
    pr = Prob(rows, cols, kind)
    t_end = float(os.environ.get("T_END", pr.T))
    t0 = time.time()
    ref, snaps = reference(pr, t_end)
    print(f"N={pr.n} {kind} reference {time.time() - t0:.1f}s norm-1={abs(np.vdot(ref, ref)) - 1:.2e}")
    ref2, _ = reference(pr, t_end, h=0.5, order=20)
    print("reference self-check (h 0.5 vs 0.25):", np.abs(ref - ref2).max())
    for name in ("strang", "s6_4", "s10_6"):
        for tau in taus:
            t0 = time.time()
            psi, out = split_run(pr, t_end, tau, name, snaps.keys())
            ph = np.vdot(ref, psi)
            errs = {k: np.abs(out[k] - snaps[k]).max() for k in sorted(out)}
            worst = max(errs.values()) if errs else float("nan")
            print(f"{name:7s} tau={tau:4.2f}  max|dpsi| end={np.abs(psi - ref).max():.3e}  worst snapshot={worst:.3e}"
                  f"  stages/ns={len(scheme(name)[1]) / tau:.0f}  ({time.time() - t0:.0f}s)")


if __name__ == "__main__":
    main()
