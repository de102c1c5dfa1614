"""Oracle, TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the Lindblad right-hand side of
``oracle/qutip_path.py:lindblad_rhs`` for two-level registers through a plain-C, matrix-free
restatement (``oracle/csrc/fast_lindblad.c``, built by ``oracle/build.py`` with gcc + OpenMP).

Why it exists: the tight reference (zvode, rtol 1e-13) of an INTERACTING 10-atom master equation -
the smallest register the split-operator row path of the product runs on - needs ~1e5 right-hand
sides of a 2^20-entry rho; the SciPy CSR products take ~0.3 s each, this one ~10 ms.

The structure (real diagonal, one |g><r| flip per atom, identical local collapse operators on every
atom: hamiltonian.py:97-124, 246-439) is READ OFF the ``OracleHamiltonian`` that
``qutip_path.build_hamiltonian`` assembled, and every use is preceded by an entry-by-entry comparison
with the SciPy right-hand side on a random non-Hermitian matrix (``check``)."""

from __future__ import annotations

import ctypes
import os

import numpy as np

from . import qutip_path as qp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfastlind.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            from . import build

            build.build()
        _lib = ctypes.CDLL(_SO)
        _lib.lind_rhs_2level.restype = None
        _lib.lind_rhs_2level.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int]
    return _lib


def local_superoperator(local_ops) -> np.ndarray:
    """4 x 4 matrix of rho_loc -> sum_c (C rho C^+ - 1/2 {C^+ C, rho}) on row-major 2 x 2 blocks."""
    S = np.zeros((4, 4), complex)
    eye = np.eye(2)
    for c in local_ops:
        c = np.asarray(c, complex)
        cdc = c.conj().T @ c
        # vec_row(A rho B) = (A kron B^T) vec_row(rho)
        S += np.kron(c, c.conj()) - 0.5 * np.kron(cdc, eye) - 0.5 * np.kron(eye, cdc.T)
    return S


class FastLindblad:
    def __init__(self, ham: qp.OracleHamiltonian):
        if ham.d != 2:
            raise ValueError("two-level registers only")
        self.ham = ham
        n = self.n = ham.n
        D = self.D = 2**n
        self.static_diag = np.ascontiguousarray(ham.static.diagonal().real)
        off = ham.static - _diag_csr(ham.static)
        if off.nnz and np.max(np.abs(off.data)) > 0:
            raise ValueError("static part is not diagonal")
        g = D - 1  # every atom in g (local index 1)
        self.dyn_diag, self.w_gr, self.w_rg = [], [], []
        for a, _ in ham.dyn_ops:
            self.dyn_diag.append(np.ascontiguousarray(a.diagonal()))
            self.w_gr.append(np.array([a[g, g ^ (1 << (n - 1 - k))] for k in range(n)], complex))
            self.w_rg.append(np.array([a[g ^ (1 << (n - 1 - k)), g] for k in range(n)], complex))
        n_spec = len(ham.collapse) // n if n else 0
        # spec-major, atom-minor (qutip_path.build_hamiltonian); atom n-1 is the least significant bit
        locs = [ham.collapse[j * n + n - 1][:2, :2].toarray() for j in range(n_spec)]
        self.S = np.ascontiguousarray(local_superoperator(locs))
        self.has_S = int(bool(locs))
        self.lib = load()

    def coefficients(self, t):
        cs = self.ham.coefficients(t)
        diag = self.static_diag.copy()
        flip = np.zeros(self.n, complex)
        for c, dd, wgr, wrg in zip(cs, self.dyn_diag, self.w_gr, self.w_rg):
            diag += 2.0 * (c * dd).real  # c a + conj(c) a^+ on the diagonal
            flip += c * wgr + np.conj(c) * np.conj(wrg)
        return diag, flip

    def __call__(self, t, y):
        y = np.ascontiguousarray(y, complex)
        out = np.empty_like(y)
        diag, flip = self.coefficients(t)
        self.lib.lind_rhs_2level(self.n, y.ctypes.data, out.ctypes.data, diag.ctypes.data, flip.ctypes.data,
                                 self.S.ctypes.data, self.has_S)
        return out

    def check(self, t=1.234, seed=5) -> float:
        y = np.random.default_rng(seed).standard_normal(2 * 4**self.n).view(complex)
        return float(np.max(np.abs(self(t, y) - qp.lindblad_rhs(self.ham)(t, y))))


def _diag_csr(m):
    import scipy.sparse as sp

    return sp.diags(m.diagonal(), format="csr")


def lindblad_rhs_fast(problem, ham=None):
    ham = ham or qp.build_hamiltonian(problem)
    return FastLindblad(ham)
