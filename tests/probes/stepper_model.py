"""NumPy model of the device stepper (CF4 Magnus; Taylor or in-place symplectic exponential)
against the tight oracle - used to choose tolerances / schemes without a GPU.

    python tests/probes/stepper_model.py N [tol] [mode]     mode = taylor | symp
"""
from __future__ import annotations

import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import blockade_radius  # noqa: E402
from oracle import qutip_path as Q  # noqa: E402
from pulser_amd import problem as P  # noqa: E402

S3 = np.sqrt(3.0)
C1, C2 = 0.5 - S3 / 6, 0.5 + S3 / 6
A1, A2 = 0.25 + S3 / 6, 0.25 - S3 / 6


def load_symp():
    txt = open(os.path.join(ROOT, "pulser_amd", "csrc", "symp_coefs.hpp")).read()
    out = []
    for mm in re.finditer(r"\{([0-9.e+-]+), ([0-9.e+-]+), (\d+),\s*\{([^}]*)\},\s*\{([^}]*)\}\}", txt):
        X, err, m = float(mm.group(1)), float(mm.group(2)), int(mm.group(3))
        a = np.array([float(v) for v in mm.group(4).split(",")])
        b = np.array([float(v) for v in mm.group(5).split(",")])
        out.append((X, err, m, a, b))
    return out


def taylor_order(rho, tol, cap=32):
    term, order = rho, 1
    while order < cap:
        term *= rho / (order + 1)
        if term <= tol:
            break
        order += 1
    return max(order, 2)


class Model:
    def __init__(self, prob):
        self.ham = Q.build_hamiltonian(prob)
        self.n = self.ham.n
        self.apps = 0

    def Hmix(self, t, h):
        """(w1, H1, H2) pieces: returns callables applying a1 H(t1) + a2 H(t2) etc. as dense-free ops."""
        ham = self.ham
        c1 = ham.coefficients(t + C1 * h)
        c2 = ham.coefficients(t + C2 * h)
        return c1, c2

    def apply_mix(self, ca, x):
        """(static * wmix + sum c op + h.c.) x with mixed coefficients `ca` (already weighted), wmix = 0.5."""
        ham = self.ham
        out = 0.5 * (ham.static @ x)
        for (a, ah), c in zip(ham.dyn_ops, ca):
            out = out + c * (a @ x) + np.conj(c) * (ah @ x)
        self.apps += 1
        return out

    def bound(self, ca):
        ham = self.ham
        D = 2 ** self.n
        diag = 0.5 * ham.static.diagonal().real
        off = 0.0
        for (a, ah), c in zip(ham.dyn_ops, ca):
            da = a.diagonal()
            if np.any(da != 0):
                diag = diag + 2 * (c * da).real
            else:
                off += abs(c) * abs(a).sum(axis=1).max() * 1.0
        lo, hi = diag.min(), diag.max()
        return 0.5 * (lo + hi), 0.5 * (hi - lo) + off

    def exp_taylor(self, ca, h, psi, tol):
        shift, bnd = self.bound(ca)
        order = taylor_order(h * bnd, tol)
        w = psi
        for j in range(order, 0, -1):
            w = psi + (h / j) * (-1j) * (self.apply_mix(ca, w) - shift * w)
        return np.exp(-1j * h * shift) * w

    def exp_symp(self, ca, h, psi, tol, table):
        shift, bnd = self.bound(ca)
        x = h * bnd
        best = None
        for nsub in range(1, 9):
            for X, err, m, a, b in table:
                if X >= x / nsub and err <= tol:
                    if best is None or nsub * m < best[0]:
                        best = (nsub * m, nsub, m, a, b)
        if best is None:
            raise RuntimeError(f"no scheme for x={x}")
        _, nsub, m, a, b = best
        hs = h / nsub
        q, p = psi.real.copy(), psi.imag.copy()

        def Hr(v):
            return (self.apply_mix(ca, v) - shift * v).real

        for _ in range(nsub):
            for i in range(m):
                q += a[i] * hs * Hr(p)
                self.apps -= 0.5
                p -= b[i] * hs * Hr(q)
                self.apps -= 0.5
            q += a[m] * hs * Hr(p)
            self.apps -= 0.5
        return np.exp(-1j * h * shift) * (q + 1j * p)

    def run(self, psi, T_ns, tol, mode, table=None):
        t = 0.0
        for k in range(T_ns):
            h = 1e-3
            c1, c2 = self.Hmix(t, h)
            for (wa, wb) in ((A1, A2), (A2, A1)):
                ca = wa * c1 + wb * c2
                psi = (self.exp_taylor(ca, h, psi, tol) if mode == "taylor"
                       else self.exp_symp(ca, h, psi, tol, table))
            t += h
        return psi


def main():
    n = int(sys.argv[1])
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-10
    mode = sys.argv[3] if len(sys.argv) > 3 else "taylor"
    kind = sys.argv[4] if len(sys.argv) > 4 else "tri"
    if kind == "tri":
        coords = P.register_coords(P.triangular_rect(2, n // 2), blockade_radius())
    else:
        coords = P.register_coords(P.square_rect(1, n), blockade_radius())
    prob = P.make_ising_problem(coords, P.anneal_samples())
    mdl = Model(prob)
    D = 2 ** mdl.n
    psi0 = np.zeros(D, complex)
    psi0[-1] = 1
    T = 3100
    t0 = time.time()
    ref = Q.sesolve(mdl.ham, psi0, [0.0, T * 1e-3], **Q.TIGHT)[-1]
    print("oracle", time.time() - t0, "s")
    table = load_symp() if mode == "symp" else None
    t0 = time.time()
    out = mdl.run(psi0.copy(), T, tol, mode, table)
    print(f"N={mdl.n} {kind} mode={mode} tol={tol:g}: max|err| = {np.max(np.abs(out - ref)):.3e}  "
          f"applications/ns = {mdl.apps / T:.2f}  ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
