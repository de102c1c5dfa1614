"""Quantum-jump (Monte-Carlo wavefunction) restatement on the CPU.

TEST INFRASTRUCTURE ONLY - imported by ``tests/`` (and nothing in the product).

What it restates
----------------
``QutipEmulator._run_solver`` hands the collapse operators to ``qutip.mcsolve``
when the noise model has stochastic noise (Solver.DEFAULT) or when
Solver.MCSOLVER is requested
(pulser-simulation/pulser_simulation/simulation.py:705-735, ``ntraj`` = 1 per
noise trajectory, = ``n_trajectories`` for the deterministic run, :843).  The
arithmetic lives in the un-vendored dependency ``qutip >= 5, < 6``
(``qutip/solver/mcsolve.py``: ``MCSolver`` / ``MCIntegrator``); its published
algorithm (Dalibard-Castin-Molmer / Dum-Zoller-Ritsch, as documented for
``mcsolve``) is:

1. draw a threshold ``r1`` ~ U(0, 1); integrate the *unnormalised* ket under
   ``H_eff = H - (i/2) sum_n C_n^dag C_n`` until ``<psi|psi> = r1``;
2. draw ``r2`` ~ U(0, 1); pick collapse operator ``n`` with probability
   ``||C_n psi||^2 / sum_m ||C_m psi||^2``; ``psi <- C_n psi / ||C_n psi||``;
3. repeat; the states reported at ``tlist`` are the normalised kets, averaged
   over trajectories as density matrices.

Parity status: qutip seeds its own bit generators (``options["seeds"]``), not
the global ``np.random`` stream, and is absent from this image - trajectory-level
parity with the reference is **unpinned** and can only be statistical.  The
statistical anchor is exact: the trajectory average converges to the
``qutip.mesolve`` state, which ``oracle/qutip_path.mesolve`` restates and which
IS pinned on the reference's golden Counters.

What is additionally fixed here (the product's ABI, ``include/rydemu.h``
``ryd_mc_solve``), so that GPU trajectories can be checked one by one:

* uniforms come from Philox4x32-10, key = the trajectory's 64-bit seed,
  counter = (jump index, 0, 0, 0): words 0-1 -> threshold, words 2-3 -> selection,
  each as ``((a >> 5) * 2**26 + (b >> 6)) / 2**53``;
* the threshold is tested at the end of every integrator step (here: the step
  grid passed in, the product: its CF4 steps, at most one sample interval) and
  the jump is applied there;
* operators are enumerated atom-major, operator-minor; the first index whose
  cumulative weight exceeds ``r2 * total`` is taken.
"""

from __future__ import annotations

from typing import Any, Sequence

import numpy as np

from . import qutip_path as qp

_M32 = 0xFFFFFFFF


def philox4x32_10(counter: Sequence[int], key: Sequence[int]) -> list[int]:
    """Philox4x32 with 10 rounds (Salmon et al., SC'11, "Parallel random
    numbers: as easy as 1, 2, 3")."""
    c = [int(x) & _M32 for x in counter]
    k0, k1 = int(key[0]) & _M32, int(key[1]) & _M32
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        hi0, lo0 = p0 >> 32, p0 & _M32
        hi1, lo1 = p1 >> 32, p1 & _M32
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + 0x9E3779B9) & _M32
        k1 = (k1 + 0xBB67AE85) & _M32
    return c


def mc_uniforms(seed: int, j: int) -> tuple[float, float]:
    """(threshold, selection) uniforms of jump number ``j``."""
    seed = int(seed)
    c = philox4x32_10([j, 0, 0, 0], [seed & _M32, (seed >> 32) & _M32])
    ut = ((c[0] >> 5) * 67108864.0 + (c[1] >> 6)) / 9007199254740992.0
    us = ((c[2] >> 5) * 67108864.0 + (c[3] >> 6)) / 9007199254740992.0
    return ut, us


def effective_rhs(ham: qp.OracleHamiltonian):
    """d psi/dt = -i H(t) psi - (1/2) sum C^dag C psi."""
    decay = None
    for c in ham.collapse:
        m = (c.conj().T @ c).tocsr()
        decay = m if decay is None else decay + m

    def rhs(t: float, y: np.ndarray) -> np.ndarray:
        out = -1j * ham.apply(t, y)
        if decay is not None:
            out = out - 0.5 * (decay @ y)
        return out

    return rhs


def mcwf_trajectory(
    ham: qp.OracleHamiltonian,
    psi0: np.ndarray,
    step_times: np.ndarray,
    eval_times: np.ndarray,
    seed: int,
    **options: Any,
) -> tuple[list[np.ndarray], list[tuple[float, int, int]]]:
    """One trajectory.  ``step_times``: the grid on which the norm threshold is
    tested (must contain every evaluation time).  Returns the normalised kets at
    ``eval_times`` and the jumps [(time, atom, operator)]."""
    opts = dict(qp.TIGHT)
    opts.update(options)
    rhs = effective_rhs(ham)
    n = ham.n
    n_ops = len(ham.collapse) // n
    step_times = np.asarray(step_times, dtype=float)
    eval_times = np.asarray(eval_times, dtype=float)
    psi = np.asarray(psi0, dtype=complex).reshape(-1).copy()
    ref = float(np.vdot(psi, psi).real)
    count = 0
    target, _ = mc_uniforms(seed, 0)
    jumps: list[tuple[float, int, int]] = []
    out: list[np.ndarray] = []
    ei = 0
    if abs(eval_times[0] - step_times[0]) < 1e-12:
        out.append(psi / np.sqrt(ref))
        ei = 1
    for i in range(len(step_times) - 1):
        t0, t1 = step_times[i], step_times[i + 1]
        psi = qp._zvode(rhs, psi, np.array([t0, t1]), opts)[-1]
        n2 = float(np.vdot(psi, psi).real)
        if n2 <= target * ref:
            # weights, atom-major / operator-minor (ham.collapse is operator-major)
            cand = []
            for a in range(n):
                for k in range(n_ops):
                    v = ham.collapse[k * n + a] @ psi
                    cand.append((a, k, v, float(np.vdot(v, v).real)))
            total = sum(c[3] for c in cand)
            if total > 0.0:
                _, us = mc_uniforms(seed, count)
                x = us * total
                cum = 0.0
                pick = None
                for c in cand:
                    cum += c[3]
                    if c[3] > 0.0 and cum > x:
                        pick = c
                        break
                if pick is None:
                    pick = [c for c in cand if c[3] > 0.0][-1]
                psi = pick[2] / np.sqrt(pick[3])
                jumps.append((float(t1), pick[0], pick[1]))
                count += 1
                target, _ = mc_uniforms(seed, count)
                ref = 1.0
                n2 = 1.0
        while ei < len(eval_times) and abs(eval_times[ei] - t1) < 1e-12:
            out.append(psi / np.sqrt(n2))
            ei += 1
    if ei != len(eval_times):
        raise ValueError("every evaluation time must be a point of the step grid")
    return out, jumps
