"""NumPy probe: CF4 (commutator-free Magnus, 2 exponentials) with steps spanning SEVERAL spline knots where
the waveform is the same polynomial across them - how long may a step be before the Magnus error shows?
Exact exponentials (high-order Taylor), so only the time-stepping error is measured.

    python tools/bigstep_probe.py ROWS COLS [tri|rect] [k ...]     (k = step length in knots / ns)
"""
from __future__ import annotations

import sys
import time

import numpy as np

from ket_split_probe import A1, A2, C1, C2, Prob, reference


def cf4_run(pr, t_end, k, kinks, guard=25, order=30):
    n = pr.n
    psi = np.zeros(1 << n, dtype=complex)
    psi[-1] = 1.0
    t = 0.0
    steps = 0
    while t < t_end - 1e-9:
        h = float(k)
        near = any(t - guard <= kk <= t + h + guard for kk in kinks)
        if near or t + h > t_end + 1e-9 or abs(t - round(t)) > 1e-9:
            h = 1.0
        for (u1, u2) in ((A1, A2), (A2, A1)):
            # norm estimate for the Taylor order: generous
            term = psi.copy()
            acc = psi.copy()
            for j in range(1, order + int(8 * h) + 1):
                term = (-1j * h / j) * (u1 * pr.apply_h(term, t + C1 * h) + u2 * pr.apply_h(term, t + C2 * h))
                acc += term
                if np.abs(term).max() < 1e-17:
                    break
            psi = acc
        t += h
        steps += 1
    return psi, steps


def main():
    rows, cols = int(sys.argv[1]), int(sys.argv[2])
    kind = sys.argv[3] if len(sys.argv) > 3 else "rect"
    ks = [int(v) for v in sys.argv[4:]] or [1, 2, 3, 4, 6]
    pr = Prob(rows, cols, kind)
    t_end = pr.T
    t0 = time.time()
    ref, _ = reference(pr, t_end)
    print(f"N={pr.n} {kind} reference {time.time() - t0:.0f}s", flush=True)
    kinks = [0, 500, 2100, 3100]
    for k in ks:
        t0 = time.time()
        psi, steps = cf4_run(pr, t_end, k, kinks)
        print(f"k={k}: steps {steps}  max|dpsi| = {np.abs(psi - ref).max():.3e}  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main()
