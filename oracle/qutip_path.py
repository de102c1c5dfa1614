"""Oracle: CPU restatement of what pulser_simulation hands to QuTiP.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  Pure NumPy/SciPy.

Input is the plain "problem" bundle of SURVEY.md Appendix C (a dict of
arrays/scalars; the product's ``pulser_amd.problem`` builds the same bundle).
No reference code is imported or copied; each function cites the reference
lines (relative to ``/root/reference``) whose behaviour it restates.

Conventions (``docs/source/conventions.md:58-73``): qudit 0 is the most
significant digit of the basis index; the local basis order is the order of
``eigenbasis`` (``["r", "g"]`` for ground-rydberg, so ``r`` = local index 0).
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Any, Callable, Mapping, Sequence

import numpy as np
import scipy.sparse as sp
from scipy.integrate import ode
from scipy.interpolate import make_interp_spline

# ---------------------------------------------------------------------------
# Local operators (hamiltonian.py:231-244)
# ---------------------------------------------------------------------------


def basis_op_matrices(eigenbasis: Sequence[str]) -> dict[str, np.ndarray]:
    """``{"I", "sigma_ab"...}`` as dense d x d arrays.

    Restates ``Hamiltonian._get_basis_op_matrices``
    (pulser-simulation/pulser_simulation/hamiltonian.py:231-244):
    ``sigma_ab = |a><b|`` with ``|a>`` the unit vector at
    ``eigenbasis.index(a)``.
    """
    d = len(eigenbasis)
    ops = {"I": np.eye(d, dtype=complex)}
    for i, a in enumerate(eigenbasis):
        for j, b in enumerate(eigenbasis):
            m = np.zeros((d, d), dtype=complex)
            m[i, j] = 1.0
            ops["sigma_" + a + b] = m
    return ops


def embed(
    n: int, d: int, factors: Mapping[int, np.ndarray]
) -> sp.csr_matrix:
    """Tensor product with ``factors[k]`` on qudit k and identity elsewhere.

    Restates ``Hamiltonian._build_operator`` for explicit qudit lists
    (hamiltonian.py:145-200, ``qutip.tensor(op_list)``): qudit 0 is the
    left-most (most significant) factor.
    """
    out = sp.identity(1, dtype=complex, format="csr")
    eye = sp.identity(d, dtype=complex, format="csr")
    for k in range(n):
        f = factors.get(k)
        out = sp.kron(
            out, eye if f is None else sp.csr_matrix(f), format="csr"
        )
    return out


def embed_global(n: int, d: int, op: np.ndarray) -> sp.csr_matrix:
    """``sum_k op_k`` (hamiltonian.py:175-179, the ``"global"`` branch)."""
    acc = sp.csr_matrix((d**n, d**n), dtype=complex)
    for k in range(n):
        acc = acc + embed(n, d, {k: op})
    return acc.tocsr()


# ---------------------------------------------------------------------------
# Knot selection + coefficients (hamiltonian.py:69-95, 333-389)
# ---------------------------------------------------------------------------


def adapt_to_sampling_rate(
    full_array: np.ndarray, sampling_rate: float, duration: int
) -> np.ndarray:
    """hamiltonian.py:87-95 - ``linspace(0, len-1, int(rate*duration), int)``."""
    idx = np.linspace(
        0, len(full_array) - 1, int(sampling_rate * duration), dtype=int
    )
    return np.asarray(full_array)[idx]


def sampling_times(duration: int, sampling_rate: float) -> np.ndarray:
    """hamiltonian.py:69-73 - knot times in microseconds."""
    return adapt_to_sampling_rate(
        np.arange(duration, dtype=np.double) / 1000, sampling_rate, duration
    )


_OP_IDS = {
    "ground-rydberg": ("sigma_gr", "sigma_rr"),
    "digital": ("sigma_hg", "sigma_gg"),
    "XY": ("sigma_ud", "sigma_dd"),
}


@dataclass
class Term:
    """One ``[operator, coefficient-knots]`` entry of the QobjEvo list.

    ``knots is None`` means a constant (coefficient 1) operator.  The full
    Hamiltonian is ``sum_k c_k(t) op_k + h.c.`` (hamiltonian.py:436-437).
    """

    op: sp.csr_matrix
    knots: np.ndarray | None
    label: str = ""


@dataclass
class OracleHamiltonian:
    n: int
    d: int
    tlist: np.ndarray
    terms: list[Term]
    collapse: list[sp.csr_matrix] = field(default_factory=list)
    # derived (filled by ``finalize``)
    static: sp.csr_matrix | None = None
    dyn_ops: list[tuple[sp.csr_matrix, sp.csr_matrix]] = field(
        default_factory=list
    )
    splines: list[Callable[[float], complex]] = field(default_factory=list)

    def finalize(self) -> "OracleHamiltonian":
        D = self.d**self.n
        static = sp.csr_matrix((D, D), dtype=complex)
        self.dyn_ops, self.splines = [], []
        for term in self.terms:
            if term.knots is None:
                # H = ham + ham.dag()  (hamiltonian.py:437)
                static = static + term.op + term.op.conj().T
            else:
                self.dyn_ops.append(
                    (term.op.tocsr(), term.op.conj().T.tocsr())
                )
                self.splines.append(coefficient_spline(self.tlist, term.knots))
        self.static = static.tocsr()
        return self

    def coefficients(self, t: float) -> np.ndarray:
        return np.array([s(t) for s in self.splines], dtype=complex)

    def matrix(self, t: float) -> sp.csr_matrix:
        """H(t) as CSR (``QobjEvo.__call__``; used by get_hamiltonian)."""
        out = self.static.copy()
        for (a, ah), c in zip(self.dyn_ops, self.coefficients(t)):
            out = out + c * a + np.conj(c) * ah
        return out.tocsr()

    def apply(self, t: float, psi: np.ndarray) -> np.ndarray:
        """H(t) @ psi for a vector or a matrix of column vectors."""
        out = self.static @ psi
        for (a, ah), c in zip(self.dyn_ops, self.coefficients(t)):
            out = out + c * (a @ psi) + np.conj(c) * (ah @ psi)
        return out


def coefficient_spline(
    tlist: np.ndarray, knots: np.ndarray
) -> Callable[[float], complex]:
    """QuTiP 5 array coefficient = cubic not-a-knot spline.

    ``qutip.QobjEvo([[op, array]], tlist=...)`` (call site hamiltonian.py:436)
    interpolates with ``scipy.interpolate.make_interp_spline(k=3)``; pinned by
    the golden Counter of test_simulation.py:981 (a natural spline gives
    581/419 instead of 572/428).
    """
    spl = make_interp_spline(tlist, np.asarray(knots, dtype=complex), k=3)
    return lambda t: complex(spl(t))


def build_hamiltonian(problem: Mapping[str, Any]) -> OracleHamiltonian:
    """Restates ``Hamiltonian._construct_hamiltonian`` (hamiltonian.py:246-439)
    and ``_build_collapse_operators`` (hamiltonian.py:97-124)."""
    n = int(problem["n_qudits"])
    eigenbasis = list(problem["eigenbasis"])
    d = len(eigenbasis)
    ops = basis_op_matrices(eigenbasis)
    duration = int(problem["duration"])
    rate = float(problem.get("sampling_rate", 1.0))
    tlist = sampling_times(duration, rate)
    bad = np.asarray(problem.get("bad_atoms", np.zeros(n, bool)), dtype=bool)
    imat = np.asarray(problem["interaction_matrix"], dtype=float)
    is_xy = problem.get("interaction_type", "ising") == "XY"
    slm_end = int(problem.get("slm_end", 0))
    slm_targets = set(problem.get("slm_targets", ()))
    basis_name = problem["basis_name"]

    def adapt(x: np.ndarray) -> np.ndarray:
        return adapt_to_sampling_rate(x, rate, duration)

    def vdw_term(i: int, j: int) -> sp.csr_matrix:  # hamiltonian.py:260-274
        u = 0.5 * imat[-1, i, j]
        return u * embed(n, d, {i: ops["sigma_rr"], j: ops["sigma_rr"]})

    def xy_term(i: int, j: int) -> sp.csr_matrix:  # hamiltonian.py:276-294
        return imat[0, i, j] * embed(
            n, d, {i: ops["sigma_ud"], j: ops["sigma_du"]}
        ) + 0.5 * imat[1, i, j] * embed(
            n, d, {i: ops["sigma_uu"], j: ops["sigma_uu"]}
        )

    def interaction_term(masked: bool = False) -> sp.csr_matrix:
        # hamiltonian.py:296-331
        D = d**n
        if masked:
            eff = n - int(bad.sum())
            for q in slm_targets:
                if not bad[q]:
                    eff -= 1
            if eff < 2:
                return sp.csr_matrix((D, D), dtype=complex)
        acc = sp.csr_matrix((D, D), dtype=complex)
        for i, j in itertools.combinations(range(n), 2):
            if bad[i] or bad[j]:
                continue
            if masked and is_xy and (i in slm_targets or j in slm_targets):
                continue
            acc = acc + (xy_term(i, j) if is_xy else vdw_term(i, j))
        return acc.tocsr()

    terms: list[Term] = []
    eff_size = n - int(bad.sum())
    if "digital" not in basis_name and eff_size > 1:  # hamiltonian.py:396
        if slm_end > 0 and is_xy:  # hamiltonian.py:399-422
            coeff = np.ones(duration - 1)
            coeff[0:slm_end] = 0
            terms.append(Term(interaction_term(), adapt(coeff), "int"))
            terms.append(
                Term(
                    interaction_term(masked=True),
                    adapt(np.logical_not(coeff).astype(int)),
                    "int_masked",
                )
            )
        else:
            terms.append(Term(interaction_term(), None, "int"))

    samples = problem["samples"]
    for addr in samples:  # hamiltonian.py:427-431
        for basis in samples[addr]:
            if not samples[addr][basis]:
                continue
            op_ids = _OP_IDS[basis]
            if addr == "Global":  # hamiltonian.py:348-365
                s = samples[addr][basis]
                coeffs = [
                    0.5 * np.asarray(s["amp"]) * np.exp(-1j * np.asarray(s["phase"])),
                    -0.5 * np.asarray(s["det"]),
                ]
                for op_id, coeff in zip(op_ids, coeffs):
                    if np.any(coeff != 0):
                        terms.append(
                            Term(
                                embed_global(n, d, ops[op_id]),
                                adapt(coeff),
                                f"G:{basis}:{op_id}",
                            )
                        )
            else:  # hamiltonian.py:366-387
                for q, s in samples[addr][basis].items():
                    coeffs = [
                        0.5 * np.asarray(s["amp"]) * np.exp(-1j * np.asarray(s["phase"])),
                        -0.5 * np.asarray(s["det"]),
                    ]
                    for coeff, op_id in zip(coeffs, op_ids):
                        if np.any(coeff != 0):
                            terms.append(
                                Term(
                                    embed(n, d, {int(q): ops[op_id]}),
                                    adapt(coeff),
                                    f"L:{basis}:{q}:{op_id}",
                                )
                            )
    if not terms:  # hamiltonian.py:433-434
        D = d**n
        terms.append(Term(sp.csr_matrix((D, D), dtype=complex), None, "zero"))

    # collapse operators, one per (spec, qudit)  (hamiltonian.py:97-124)
    collapse: list[sp.csr_matrix] = []
    paulis = problem.get("depolarizing_pauli_2ds", {})
    for coeff, cop in problem.get("collapse_ops", []):
        if isinstance(cop, str):
            if cop not in ops:
                local = sum(coeff * pc * ops[po] for pc, po in paulis[cop])
            else:
                local = coeff * ops[cop]
        else:
            local = coeff * np.asarray(cop, dtype=complex)
        for k in range(n):
            collapse.append(embed(n, d, {k: local}))

    return OracleHamiltonian(n, d, tlist, terms, collapse).finalize()


# ---------------------------------------------------------------------------
# Solvers (simulation.py:689-797 -> qutip.sesolve / mesolve)
# ---------------------------------------------------------------------------

QUTIP_DEFAULTS = dict(atol=1e-8, rtol=1e-6, order=12, method="adams")


def _zvode(
    rhs: Callable[[float, np.ndarray], np.ndarray],
    y0: np.ndarray,
    eval_times: np.ndarray,
    options: Mapping[str, Any],
    counter: list[int] | None = None,
) -> list[np.ndarray]:
    """QuTiP 5 ``IntegratorScipyAdams``: scipy ``zvode`` driven to each t."""
    opt = dict(QUTIP_DEFAULTS)
    opt.update({k: v for k, v in options.items() if v is not None})
    method = opt.pop("method", "adams")

    def f(t: float, y: np.ndarray) -> np.ndarray:
        if counter is not None:
            counter[0] += 1
        return rhs(t, y)

    r = ode(f)
    r.set_integrator(
        "zvode",
        method=method,
        atol=opt["atol"],
        rtol=opt["rtol"],
        order=opt.get("order", 12),
        nsteps=int(opt.get("nsteps", 2500)),
        max_step=float(opt.get("max_step", 0.0)),
        first_step=float(opt.get("first_step", 0.0)),
        min_step=float(opt.get("min_step", 0.0)),
    )
    t0 = float(eval_times[0])
    r.set_initial_value(np.asarray(y0, dtype=complex), t0)
    out = [np.array(y0, dtype=complex)]
    for t in eval_times[1:]:
        r.integrate(float(t))
        if not r.successful():  # pragma: no cover
            raise RuntimeError(f"zvode failed at t={t}")
        out.append(r.y.copy())
    return out


def sesolve(
    ham: OracleHamiltonian,
    psi0: np.ndarray,
    eval_times: np.ndarray,
    counter: list[int] | None = None,
    **options: Any,
) -> list[np.ndarray]:
    """``qutip.sesolve(H, psi0, tlist, options)`` (simulation.py:705, 729-735);
    ``normalize_output=False`` (simulation.py:720-721)."""

    def rhs(t: float, y: np.ndarray) -> np.ndarray:
        return -1j * ham.apply(t, y)

    return _zvode(rhs, psi0, np.asarray(eval_times, float), options, counter)


def lindblad_rhs(
    ham: OracleHamiltonian,
) -> Callable[[float, np.ndarray], np.ndarray]:
    """d rho/dt = -i[H,rho] + sum_c (C rho C^+ - 1/2 {C^+ C, rho})."""
    D = ham.d**ham.n
    cs = ham.collapse
    cdc = sp.csr_matrix((D, D), dtype=complex)
    for c in cs:
        cdc = cdc + c.conj().T @ c
    cdc = cdc.tocsr()
    cs_h = [c.conj().T.tocsr() for c in cs]

    def rhs(t: float, y: np.ndarray) -> np.ndarray:
        rho = y.reshape(D, D)
        h_rho = ham.apply(t, rho)
        # rho H = (H rho^+)^+ ; do not assume rho Hermitian
        rho_h = ham.apply(t, rho.conj().T).conj().T
        out = -1j * (h_rho - rho_h)
        anti = cdc @ rho
        out -= 0.5 * (anti + (cdc @ rho.conj().T).conj().T)
        for c, ch in zip(cs, cs_h):
            # C rho C^+ = (C (C rho)^+)^+
            out += (c @ (c @ rho).conj().T).conj().T
        return out.ravel()

    return rhs


def mesolve(
    ham: OracleHamiltonian,
    rho0: np.ndarray,
    eval_times: np.ndarray,
    counter: list[int] | None = None,
    **options: Any,
) -> list[np.ndarray]:
    """``qutip.mesolve(H, psi0, tlist, c_ops=..., options)``
    (simulation.py:707-735).  ``rho0`` may be a ket (converted to |psi><psi|).
    Returns D x D matrices."""
    D = ham.d**ham.n
    rho0 = np.asarray(rho0, dtype=complex)
    if rho0.ndim == 1 or (rho0.ndim == 2 and rho0.shape[1] == 1):
        v = rho0.reshape(-1)
        rho0 = np.outer(v, v.conj())
    ys = _zvode(
        lindblad_rhs(ham),
        rho0.ravel(),
        np.asarray(eval_times, float),
        options,
        counter,
    )
    return [y.reshape(D, D) for y in ys]


TIGHT = dict(atol=1e-15, rtol=1e-13, nsteps=100_000_000)


def liouvillian(ham: OracleHamiltonian, t: float) -> sp.csr_matrix:
    """Explicit column-stacked superoperator (what qutip.liouvillian builds);
    only for cross-checking ``lindblad_rhs`` at tiny N."""
    D = ham.d**ham.n
    eye = sp.identity(D, dtype=complex, format="csr")
    h = ham.matrix(t)
    L = -1j * (sp.kron(eye, h) - sp.kron(h.T, eye))
    for c in ham.collapse:
        cdc = c.conj().T @ c
        L = L + sp.kron(c.conj(), c) - 0.5 * sp.kron(eye, cdc) - 0.5 * sp.kron(cdc.T, eye)
    return L.tocsr()


# ---------------------------------------------------------------------------
# Default solver options (simulation.py:663-687, 768-780)
# ---------------------------------------------------------------------------


def min_variation(amp: np.ndarray, det: np.ndarray) -> int:
    """``QutipEmulator._get_min_variation`` (simulation.py:663-687)."""
    end_point = len(amp) - 1
    mins = []
    for sample in (np.asarray(amp), np.asarray(det)):
        mins.append(
            int(
                np.min(
                    np.diff(
                        np.nonzero(np.diff(sample)),
                        prepend=-1,
                        append=end_point,
                    )
                )
            )
        )
    return min(mins)


def default_options(
    channel_samples: Sequence[tuple[np.ndarray, np.ndarray]], tot_duration: int
) -> dict[str, float]:
    """``_validate_options`` defaults (simulation.py:768-780); ``channel_samples``
    = (amp, det) of every channel of the *extended* samples."""
    max_step = min(min_variation(a, dt) for a, dt in channel_samples) / 1000
    nsteps = max(1000, tot_duration // max_step)
    return {"max_step": max_step, "nsteps": nsteps}


def all_ground_state(n: int, eigenbasis: Sequence[str], xy: bool = False) -> np.ndarray:
    """``set_initial_state("all-ground")`` (simulation.py:498-505)."""
    d = len(eigenbasis)
    loc = list(eigenbasis).index("u" if xy else "g")
    idx = 0
    for _ in range(n):
        idx = idx * d + loc
    psi = np.zeros(d**n, dtype=complex)
    psi[idx] = 1.0
    return psi
