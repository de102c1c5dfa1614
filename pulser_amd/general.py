"""General (any local dimension) lowering: a term list ``G(t) = sum_t coef_t(t) A_t``.

The tuned kernels cover the 2-level Ising problems that dominate the workload
sizes of BASELINE.json.  Everything else the reference's ``Hamiltonian`` can
express - the 3-level ``"all"`` basis, leakage (``*_with_error``, d = 3/4), XY
mode incl. the SLM-mask switching terms, and arbitrary ``eff_noise`` collapse
operators - is lowered here and integrated on the GPU by the same CF4/Taylor
stepper through the ``ryd_general_*`` entry points.

``matrix_free=True`` (the default above 4096 amplitudes, i.e. beyond the one-launch kernel): every operator of the reference's lists is a sum
of one- and two-site operators or a diagonal, and is handed over AS THAT - a
d x d or d^2 x d^2 matrix with the digit strides and weights of the sites it acts
on (``ryd_general_add_local_term``), or a dense diagonal
(``ryd_general_add_diag_term``); the kernel decodes the digits of a row and
gathers - site by site since round 3: all terms acting on a site are added into one small matrix per
exponential (``k_gen_apply_sites``).  Nothing of size d^N x nnz is built on the host or stored on the device.
``matrix_free=False`` (the default below that): the same generator as explicit CSR
matrices - faster for the small systems of the reference's tests, and the cross-check.

Term structure restated from
pulser-simulation/pulser_simulation/hamiltonian.py:246-439 (Hamiltonian) and
:97-124 (collapse operators); the Liouvillian uses row-major vec(rho):
``H rho -> (H (x) I) vec``, ``rho H -> (I (x) H^T) vec``.
"""

from __future__ import annotations

import itertools
import os
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
    # matrix-free terms (None entries of row_ptr / col_idx / values mark them): per term None (CSR) or
    # ("diag", complex128[dim]) or ("local", d, n_per, int64[G, n_per] strides, float64[G] weights,
    #  int32[nnz] rows, int32[nnz] cols, complex128[nnz] vals)
    free: list | None = None


@dataclass
class _LocalSum:
    """sum_g w_g embed(M on the sites of group g); sites are digit positions of the evolved vector."""

    mat: np.ndarray  # (d^p, d^p)
    groups: list  # [(sites tuple of length p, weight)]

    def dagger(self) -> "_LocalSum":
        return _LocalSum(self.mat.conj().T, self.groups)


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


MATRIX_FREE_FROM = 4097  # evolved-vector length from which the matrix-free terms are the default


def lower_general(problem: Mapping[str, Any], mesolve: bool, matrix_free: bool | None = None) -> GeneralTables:
    """``matrix_free=None``: explicit CSR terms for the systems the one-launch kernel holds (<= 4096 entries:
    every case of the reference's tests; precomputed indices beat the digit decode there), matrix-free terms
    above: the site-fused application measures 11 ms against 16 ms (CSR) and 34 ms (term by term, round 2)
    on a 40-ns solve at 19 683 amplitudes, and nothing of the operators' size is built or stored
    (``scipy.sparse.kron``: 40x the lowering time already at 3^9)."""
    if matrix_free is None:
        d, n = len(problem["eigenbasis"]), int(problem["n_qudits"])
        # RYD_GENERAL_MATRIX_FREE=1: every general-path solve on the matrix-free terms (how the end-to-end
        # golden tests are run through them: RYD_GENERAL_MATRIX_FREE=1 pytest tests/test_gpu_emulator.py -m gpu)
        matrix_free = (d ** (2 * n if mesolve else n) >= MATRIX_FREE_FROM
                       or os.environ.get("RYD_GENERAL_MATRIX_FREE") == "1")
    if matrix_free:
        tables = _lower_matrix_free(problem, mesolve)
        if tables is not None:
            return tables
    return _lower_csr(problem, mesolve)


def _spline_pp(tknots: np.ndarray, series_knots: list) -> np.ndarray:
    pp = np.empty((len(series_knots), len(tknots) - 1, 4), dtype=np.complex128)
    for i, kn in enumerate(series_knots):
        pp[i] = np.transpose(CubicSpline(tknots, kn, bc_type="not-a-knot").c, (1, 0))
    return pp


def _tighten_static_norms(terms: list, dim: int) -> list:
    """The step-size bound of the library is sum_t |coef_t| ||A_t||_inf.  For the TIME-INDEPENDENT terms (the interaction)
    the sum of the separate norms over-counts: ``op`` and ``op^dag`` of an exchange pair never fill the same row, a pair
    term only has entries on rows whose two digits match its local rows (an XY register of 12 atoms: 4 x sum |C3 / r^3|
    against a largest row sum of ~ 1/4 of that).  Their exact joint bound max_row sum_t sum_g |w_g| rowsum|M_t|[R_g(row)]
    costs dim x groups operations here, once; the static terms' norms are scaled so that they add up to it (same ABI:
    ``row_norm`` per term).  The Taylor order and the number of sub-steps follow the bound, the truncation criterion
    (remainder <= tol for rho = h x bound) is unchanged."""
    static = [i for i, t in enumerate(terms) if t[1] < 0 and t[3] > 0.0]
    work = sum(len(terms[i][0][4]) if terms[i][0][0] == "local" else 1 for i in static)
    if len(static) < 2 or work * dim > 2e8:
        return terms
    idx = np.arange(dim, dtype=np.int64)
    r = np.zeros(dim)
    for i in static:
        pay = terms[i][0]
        if pay[0] == "diag":
            r += np.abs(pay[1])
            continue
        _, d, p, st, w, rr, cc, vals = pay
        rowabs = np.zeros(d**p)
        np.add.at(rowabs, rr, np.abs(vals))
        dig: dict[int, np.ndarray] = {}
        for g in range(len(w)):
            for q in range(p):
                sq = int(st[g, q])
                if sq not in dig:
                    dig[sq] = (idx // sq) % d
            R = dig[int(st[g, 0])] if p == 1 else dig[int(st[g, 0])] * d + dig[int(st[g, 1])]
            r += abs(float(w[g])) * rowabs[R]
    exact, crude = float(r.max()), float(sum(terms[i][3] for i in static))
    if not (0.0 < exact < crude):
        return terms
    f = exact / crude
    return [(t[0], t[1], t[2], t[3] * f) if i in static else t for i, t in enumerate(terms)]


def _lower_matrix_free(problem: Mapping[str, Any], mesolve: bool) -> GeneralTables | None:
    """The term list of :func:`_lower_csr` without materialising any operator: Hamiltonian terms
    (hamiltonian.py:246-439) as site sums / pair sums / diagonals, the Liouvillian
    ``-i (H (x) I - I (x) H^T)`` as the same sums on the row and the column digits of row-major
    vec(rho), the dissipator (hamiltonian.py:97-124) as ONE d^2 x d^2 superoperator on the digit
    pairs (row_k, col_k)."""
    n = int(problem["n_qudits"])
    eigenbasis = list(problem["eigenbasis"])
    d = len(eigenbasis)
    D = d**n
    n_dig = 2 * n if mesolve else n
    dim = D * D if mesolve else D
    if dim > (1 << 26):
        raise NotImplementedError(f"The general (multi-level / XY) path evolves vectors of at most 2^26 "
                                  f"entries; this system needs {dim}.")
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
    stride = [d ** (n_dig - 1 - p) for p in range(n_dig)]  # digit position -> stride

    def adapt(x: np.ndarray) -> np.ndarray:
        return adapt_to_sampling_rate(x, rate, duration)

    def digits(pos: int) -> np.ndarray:
        return (np.arange(dim, dtype=np.int64) // stride[pos]) % d

    # -- Hamiltonian as (object, knots | None); object = _LocalSum or ("diag", real vector over D) --
    h_terms: list[tuple[Any, np.ndarray | None]] = []

    def interaction(masked: bool = False) -> list:
        out: list = []
        if masked:
            eff = n - int(bad.sum()) - sum(1 for q in slm_targets if not bad[q])
            if eff < 2:
                return out
        pairs = [(i, j) for i, j in itertools.combinations(range(n), 2)
                 if not (bad[i] or bad[j]) and not (masked and is_xy and (i in slm_targets or j in slm_targets))]
        if not pairs:
            return out
        if is_xy:  # hamiltonian.py:276-294: op = sum_ij imat0 ud_i du_j + 0.5 imat1 uu_i uu_j  (H = op + op^dag)
            out.append(_LocalSum(np.kron(ops["sigma_ud"], ops["sigma_du"]), [((i, j), imat[0, i, j]) for i, j in pairs]))
            if np.any([imat[1, i, j] != 0 for i, j in pairs]):
                out.append(_LocalSum(np.kron(ops["sigma_uu"], ops["sigma_uu"]),
                                     [((i, j), 0.5 * imat[1, i, j]) for i, j in pairs]))
        else:  # :260-274: 0.5 U_ij rr_i rr_j, doubled by op + op^dag: a diagonal
            r = eigenbasis.index("r")
            nD = [(np.arange(D, dtype=np.int64) // d ** (n - 1 - k)) % d == r for k in range(n)]
            e = np.zeros(D)
            for i, j in pairs:
                e += 0.5 * imat[-1, i, j] * (nD[i] & nD[j])
            out.append(("diag", e))
        return out

    if "digital" not in basis_name and (n - int(bad.sum())) > 1:  # :393-424
        if slm_end > 0 and is_xy:
            coeff = np.ones(duration - 1)
            coeff[0:slm_end] = 0
            h_terms += [(o, adapt(coeff)) for o in interaction()]
            h_terms += [(o, adapt(np.logical_not(coeff).astype(int))) for o in interaction(masked=True)]
        else:
            h_terms += [(o, None) for o in interaction()]
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
                    sites = range(n) if q is None else [q]
                    h_terms.append((_LocalSum(ops[op_id], [((k,), 1.0) for k in sites]), adapt(coeff)))

    # -- generator terms: (payload, series index | -1, conj) --
    terms: list[tuple[tuple, int, int, float]] = []
    series_knots: list[np.ndarray] = []

    def local_payload(ls: _LocalSum, factor: complex, col_side: bool) -> tuple[tuple, float]:
        """factor * sum_g w_g embed(M) on the row digits, or factor * embed(M^T) on the column digits."""
        m = ls.mat.T if col_side else ls.mat
        p = len(ls.groups[0][0])
        rr, cc = np.nonzero(m)
        if len(rr) == 0:
            return (), 0.0
        shift = n if col_side else 0
        st = np.array([[stride[site + shift] for site in sites] for sites, _ in ls.groups], dtype=np.int64)
        w = np.array([wt for _, wt in ls.groups], dtype=np.float64)
        vals = (factor * m[rr, cc]).astype(np.complex128)
        norm = float(np.abs(w).sum() * np.abs(factor * m).sum(axis=1).max())
        return ("local", d, p, st, w, rr.astype(np.int32), cc.astype(np.int32), vals), norm

    def add_generator(obj: Any, series: int, conj: int, dagger: bool) -> None:
        """-i obj (kets) or -i [obj, .] (vec rho); `dagger`: of obj^dag."""
        if isinstance(obj, tuple):  # real diagonal: its own adjoint
            e = obj[1]
            if mesolve:
                v = -1j * (np.repeat(e, D) - np.tile(e, D))
            else:
                v = -1j * e
            terms.append((("diag", np.ascontiguousarray(v, dtype=np.complex128)), series, conj, float(np.abs(v).max())))
            return
        ls = obj.dagger() if dagger else obj
        pay, norm = local_payload(ls, -1j, False)
        if pay:
            terms.append((pay, series, conj, norm))
        if mesolve:
            pay, norm = local_payload(ls, 1j, True)
            if pay:
                terms.append((pay, series, conj, norm))

    for obj, knots in h_terms:
        if knots is None:
            si = -1
        else:
            series_knots.append(np.asarray(knots, dtype=complex))
            si = len(series_knots) - 1
        if isinstance(obj, tuple):
            # op + op^dag of a real diagonal with a (real) coefficient: 2 * diag, coefficient conj-invariant
            add_generator(("diag", 2.0 * obj[1]), si, 0, False)
        else:
            add_generator(obj, si, 0, False)
            add_generator(obj, si, 1, True)

    if mesolve:
        # D[rho] = sum_c  c rho c^dag - 1/2 {c^dag c, rho}: per atom the same d^2 x d^2 superoperator on the
        # digit pair (row_k, col_k): L (x) conj(L) - 1/2 (L^dag L (x) I) - 1/2 (I (x) (L^dag L)^T)
        paulis = problem.get("depolarizing_pauli_2ds", {})
        sup = np.zeros((d * d, d * d), dtype=complex)
        eye_d = np.eye(d)
        for coeff, cop in problem.get("collapse_ops", []):
            if isinstance(cop, str):
                local = coeff * ops[cop] if cop in ops else sum(coeff * pc * ops[po] for pc, po in paulis[cop])
            else:
                local = coeff * np.asarray(cop, dtype=complex)
            ldl = local.conj().T @ local
            sup += np.kron(local, local.conj()) - 0.5 * np.kron(ldl, eye_d) - 0.5 * np.kron(eye_d, ldl.T)
        if np.any(sup != 0):
            ls = _LocalSum(sup, [((k, n + k), 1.0) for k in range(n)])
            rr, cc = np.nonzero(sup)
            st = np.array([[stride[k], stride[n + k]] for k in range(n)], dtype=np.int64)
            pay = ("local", d, 2, st, np.ones(n), rr.astype(np.int32), cc.astype(np.int32),
                   sup[rr, cc].astype(np.complex128))
            terms.append((pay, -1, 0, float(n * np.abs(sup).sum(axis=1).max())))
            del ls
    if len(terms) > 96:  # MAX_GEN_TERMS of the library: fall back to merged CSR terms
        return None
    terms = _tighten_static_norms(terms, dim)
    if not terms:
        terms.append((("diag", np.zeros(dim, dtype=np.complex128)), -1, 0, 0.0))
    if not series_knots:
        series_knots.append(np.zeros(len(tknots), dtype=complex))
    nt = len(terms)
    return GeneralTables(
        dim=dim, n_qudits=n, local_dim=d, is_density=mesolve,
        tknots=np.ascontiguousarray(tknots, dtype=np.float64), pp=np.ascontiguousarray(_spline_pp(tknots, series_knots)),
        row_ptr=[None] * nt, col_idx=[None] * nt, values=[None] * nt,
        series=np.asarray([t[1] for t in terms], dtype=np.int32), conj=np.asarray([t[2] for t in terms], dtype=np.int32),
        scale=np.ones(nt, dtype=np.complex128), row_norm=np.asarray([t[3] for t in terms], dtype=np.float64),
        free=[t[0] for t in terms],
    )


def dense_generator(tables: GeneralTables, coefs: Sequence[complex]) -> np.ndarray:
    """sum_t coefs[t] A_t as a dense matrix (tests: the matrix-free terms against the CSR ones)."""
    out = np.zeros((tables.dim, tables.dim), dtype=complex)
    idx = np.arange(tables.dim, dtype=np.int64)
    for t in range(len(tables.values)):
        free = tables.free[t] if tables.free is not None else None
        if free is None:
            m = sp.csr_matrix((tables.values[t], tables.col_idx[t], tables.row_ptr[t]), shape=out.shape)
            out += coefs[t] * m.toarray()
        elif free[0] == "diag":
            out[idx, idx] += coefs[t] * free[1]
        else:
            _, d, p, st, w, rr, cc, vals = free
            for g in range(len(w)):
                a = (idx // st[g, 0]) % d
                if p == 2:
                    b = (idx // st[g, 1]) % d
                    R = a * d + b
                else:
                    b = 0
                    R = a
                for e in range(len(vals)):
                    rows = idx[R == rr[e]]
                    if p == 2:
                        cols = rows + (cc[e] // d - a[rows]) * st[g, 0] + (cc[e] % d - b[rows]) * st[g, 1]
                    else:
                        cols = rows + (cc[e] - a[rows]) * st[g, 0]
                    np.add.at(out, (rows, cols), coefs[t] * w[g] * vals[e])
    return out


def _lower_csr(problem: Mapping[str, Any], mesolve: bool) -> GeneralTables:
    n = int(problem["n_qudits"])
    eigenbasis = list(problem["eigenbasis"])
    d = len(eigenbasis)
    D = d**n
    dim = D * D if mesolve else D
    if dim > MAX_DIM:
        raise NotImplementedError(
            f"The explicit-CSR lowering materialises sparse operators of size "
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
    pp = _spline_pp(tknots, series_knots)
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
