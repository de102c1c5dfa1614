"""General (any local dimension) lowering: explicit sparse terms for small systems.

The tuned matrix-free kernels cover the 2-level Ising problems that dominate
the workload sizes of BASELINE.json.  Everything else the reference's
``Hamiltonian`` can express - the 3-level ``"all"`` basis, leakage
(``*_with_error``, d = 3/4), XY mode incl. the SLM-mask switching terms, and
arbitrary ``eff_noise`` collapse operators - is lowered here to an explicit
list ``G(t) = sum_t coef_t(t) A_t`` of CSR matrices with spline coefficients
and integrated on the GPU by the same CF4/Taylor stepper through the
``ryd_general_*`` entry points (a row-per-thread CSR kernel; these systems are
small: d**N, or d**2N for the Liouvillian).

Term structure restated from
pulser-simulation/pulser_simulation/hamiltonian.py:246-439 (Hamiltonian) and
:97-124 (collapse operators); the Liouvillian uses row-major vec(rho):
``H rho -> (H (x) I) vec``, ``rho H -> (I (x) H^T) vec``.
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Any, Mapping, Sequence

import numpy as np
import scipy.sparse as sp
from scipy.interpolate import CubicSpline

from .terms import adapt_to_sampling_rate, sampling_times

_OP_IDS = {
    "ground-rydberg": ("sigma_gr", "sigma_rr"),
    "digital": ("sigma_hg", "sigma_gg"),
    "XY": ("sigma_ud", "sigma_dd"),
}
MAX_DIM = 1 << 22  # vector length the general path accepts


@dataclass
class GeneralTables:
    dim: int  # length of the evolved vector (d**N or d**2N)
    n_qudits: int
    local_dim: int
    is_density: bool
    tknots: np.ndarray
    pp: np.ndarray  # complex128[n_series][n_knots-1][4]
    # per term: CSR + (series index or -1 for static, complex scale, conj flag)
    row_ptr: list[np.ndarray]
    col_idx: list[np.ndarray]
    values: list[np.ndarray]
    series: np.ndarray  # int32[n_terms]
    conj: np.ndarray  # int32[n_terms]
    scale: np.ndarray  # complex128[n_terms]
    row_norm: np.ndarray  # float64[n_terms]: max abs row sum


def _local_ops(eigenbasis: Sequence[str]) -> dict[str, np.ndarray]:
    d = len(eigenbasis)
    ops = {"I": np.eye(d, dtype=complex)}
    for i, a in enumerate(eigenbasis):
        for j, b in enumerate(eigenbasis):
            m = np.zeros((d, d), dtype=complex)
            m[i, j] = 1.0
            ops["sigma_" + a + b] = m
    return ops


def _embed(n: int, d: int, factors: Mapping[int, np.ndarray]) -> sp.csr_matrix:
    out = sp.identity(1, dtype=complex, format="csr")
    eye = sp.identity(d, dtype=complex, format="csr")
    for k in range(n):
        f = factors.get(k)
        out = sp.kron(out, eye if f is None else sp.csr_matrix(f), format="csr")
    return out


def lower_general(problem: Mapping[str, Any], mesolve: bool) -> GeneralTables:
    n = int(problem["n_qudits"])
    eigenbasis = list(problem["eigenbasis"])
    d = len(eigenbasis)
    D = d**n
    dim = D * D if mesolve else D
    if dim > MAX_DIM:
        raise NotImplementedError(
            f"The general (multi-level / XY) path materialises sparse operators of size "
            f"{dim}; systems beyond {MAX_DIM} entries are not supported."
        )
    ops = _local_ops(eigenbasis)
    duration = int(problem["duration"])
    rate = float(problem.get("sampling_rate", 1.0))
    tknots = sampling_times(duration, rate)
    bad = np.asarray(problem.get("bad_atoms", np.zeros(n, bool)), dtype=bool)
    imat = np.asarray(problem["interaction_matrix"], dtype=float)
    is_xy = problem.get("interaction_type", "ising") == "XY"
    slm_end = int(problem.get("slm_end", 0))
    slm_targets = set(problem.get("slm_targets", ()))
    basis_name = problem["basis_name"]

    def adapt(x: np.ndarray) -> np.ndarray:
        return adapt_to_sampling_rate(x, rate, duration)

    def interaction(masked: bool = False) -> sp.csr_matrix:  # hamiltonian.py:296-331
        acc = sp.csr_matrix((D, D), dtype=complex)
        if masked:
            eff = n - int(bad.sum()) - sum(1 for q in slm_targets if not bad[q])
            if eff < 2:
                return acc
        for i, j in itertools.combinations(range(n), 2):
            if bad[i] or bad[j]:
                continue
            if masked and is_xy and (i in slm_targets or j in slm_targets):
                continue
            if is_xy:  # :276-294
                acc = acc + imat[0, i, j] * _embed(n, d, {i: ops["sigma_ud"], j: ops["sigma_du"]}) \
                    + 0.5 * imat[1, i, j] * _embed(n, d, {i: ops["sigma_uu"], j: ops["sigma_uu"]})
            else:  # :260-274
                acc = acc + 0.5 * imat[-1, i, j] * _embed(n, d, {i: ops["sigma_rr"], j: ops["sigma_rr"]})
        return acc.tocsr()

    # Hamiltonian terms [(operator, knots or None)]; H = sum c_k op_k + h.c.
    h_terms: list[tuple[sp.csr_matrix, np.ndarray | None]] = []
    if "digital" not in basis_name and (n - int(bad.sum())) > 1:  # :393-424
        if slm_end > 0 and is_xy:
            coeff = np.ones(duration - 1)
            coeff[0:slm_end] = 0
            h_terms.append((interaction(), adapt(coeff)))
            h_terms.append((interaction(masked=True), adapt(np.logical_not(coeff).astype(int))))
        else:
            h_terms.append((interaction(), None))
    samples = problem["samples"]
    for addr in samples:  # :427-431
        for basis, s in samples[addr].items():
            if not s:
                continue
            op_ids = _OP_IDS[basis]
            entries = [(None, s)] if addr == "Global" else [(int(q), sq) for q, sq in s.items()]
            for q, sq in entries:
                coeffs = [0.5 * np.asarray(sq["amp"]) * np.exp(-1j * np.asarray(sq["phase"])),
                          -0.5 * np.asarray(sq["det"])]
                for op_id, coeff in zip(op_ids, coeffs):
                    if not np.any(coeff != 0):
                        continue
                    if q is None:
                        op = sum(_embed(n, d, {k: ops[op_id]}) for k in range(n))
                    else:
                        op = _embed(n, d, {q: ops[op_id]})
                    h_terms.append((sp.csr_matrix(op), adapt(coeff)))

    # collapse operators (hamiltonian.py:97-124), one per (spec, atom)
    collapse: list[sp.csr_matrix] = []
    paulis = problem.get("depolarizing_pauli_2ds", {})
    for coeff, cop in problem.get("collapse_ops", []):
        if isinstance(cop, str):
            local = coeff * ops[cop] if cop in ops else sum(coeff * pc * ops[po] for pc, po in paulis[cop])
        else:
            local = coeff * np.asarray(cop, dtype=complex)
        for k in range(n):
            collapse.append(_embed(n, d, {k: local}))

    eye = sp.identity(D, dtype=complex, format="csr")

    def gen(op: sp.csr_matrix) -> sp.csr_matrix:
        """-i op (kets) or -i [op, .] (row-major vec rho)."""
        if not mesolve:
            return (-1j * op).tocsr()
        return (-1j * (sp.kron(op, eye) - sp.kron(eye, op.T))).tocsr()

    mats: list[sp.csr_matrix] = []
    series_knots: list[np.ndarray] = []
    series_idx: list[int] = []
    conj: list[int] = []
    static = sp.csr_matrix((dim, dim), dtype=complex)
    for op, knots in h_terms:
        opd = op.conj().T.tocsr()
        if knots is None:
            static = static + gen(op) + gen(opd)
            continue
        series_knots.append(np.asarray(knots, dtype=complex))
        si = len(series_knots) - 1
        mats.append(gen(op)); series_idx.append(si); conj.append(0)
        mats.append(gen(opd)); series_idx.append(si); conj.append(1)
    if mesolve:
        for c in collapse:
            cdc = (c.conj().T @ c).tocsr()
            static = static + sp.kron(c, c.conj()) - 0.5 * sp.kron(cdc, eye) - 0.5 * sp.kron(eye, cdc.T)
    static = static.tocsr()
    static.eliminate_zeros()
    if static.nnz:
        mats.append(static); series_idx.append(-1); conj.append(0)
    if not mats:
        mats.append(sp.csr_matrix((dim, dim), dtype=complex)); series_idx.append(-1); conj.append(0)
    if not series_knots:
        series_knots.append(np.zeros(len(tknots), dtype=complex))
    pp = np.empty((len(series_knots), len(tknots) - 1, 4), dtype=np.complex128)
    for i, kn in enumerate(series_knots):
        pp[i] = np.transpose(CubicSpline(tknots, kn, bc_type="not-a-knot").c, (1, 0))
    row_ptr, col_idx, values, norms = [], [], [], []
    for m in mats:
        m = m.tocsr()
        m.sort_indices()
        row_ptr.append(np.ascontiguousarray(m.indptr, dtype=np.int32))
        col_idx.append(np.ascontiguousarray(m.indices, dtype=np.int32))
        values.append(np.ascontiguousarray(m.data, dtype=np.complex128))
        norms.append(float(abs(m).sum(axis=1).max()) if m.nnz else 0.0)
    return GeneralTables(
        dim=dim, n_qudits=n, local_dim=d, is_density=mesolve,
        tknots=np.ascontiguousarray(tknots, dtype=np.float64), pp=np.ascontiguousarray(pp),
        row_ptr=row_ptr, col_idx=col_idx, values=values,
        series=np.asarray(series_idx, dtype=np.int32), conj=np.asarray(conj, dtype=np.int32),
        scale=np.ones(len(mats), dtype=np.complex128), row_norm=np.asarray(norms, dtype=np.float64),
    )
