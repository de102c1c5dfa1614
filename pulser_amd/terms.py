"""Lower problem bundles to the device tables of ``librydemu``.

Restates, matrix-free, what ``Hamiltonian._construct_hamiltonian`` builds as a
``qutip.QobjEvo`` (pulser-simulation/pulser_simulation/hamiltonian.py:246-439):

* knots = samples at ``linspace(0, len-1, int(rate*duration), dtype=int)``
  (hamiltonian.py:87-95), knot times ``arange(duration)/1000`` us (:69-73);
* per global channel / per local qubit the coefficient ``0.5*amp*exp(-i*phase)``
  on ``sigma_gr`` (+ h.c.) and ``-0.5*det`` on ``sigma_rr`` (+ h.c. = ``-det n``)
  (hamiltonian.py:340-352, 370-375);
* the static ``sum_{i<j} U_ij n_i n_j`` from ``interaction_matrix[-1]``
  (hamiltonian.py:260-274, 308-331), absent for a digital-only basis or fewer
  than two good atoms (:393-396);
* QuTiP's array coefficients are cubic *not-a-knot* splines; the
  piecewise polynomials are computed here on the host and uploaded.

and the collapse operators of ``_build_collapse_operators``
(hamiltonian.py:97-124) as the 4x4 local superoperator
``S = sum_c C (x) conj(C) - 1/2 (C^+C (x) I) - 1/2 (I (x) (C^+C)^T)`` on the
digit pair (a_k, b_k) of rho.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Mapping, Sequence

import numpy as np
from scipy.interpolate import CubicSpline

SUPPORTED_BASES = ("ground-rydberg", "digital")


def adapt_to_sampling_rate(
    full_array: np.ndarray, sampling_rate: float, duration: int
) -> np.ndarray:
    """``Hamiltonian._adapt_to_sampling_rate`` (hamiltonian.py:87-95)."""
    idx = np.linspace(
        0, len(full_array) - 1, int(sampling_rate * duration), dtype=int
    )
    return np.asarray(full_array)[idx]


def sampling_times(duration: int, sampling_rate: float) -> np.ndarray:
    """Knot times in us (hamiltonian.py:69-73)."""
    return adapt_to_sampling_rate(
        np.arange(duration, dtype=np.double) / 1000, sampling_rate, duration
    )


@dataclass
class DeviceTables:
    """Everything ``ryd_set_*`` needs, as host arrays."""

    n_qubits: int
    batch: int
    tknots: np.ndarray  # float64[n_knots]
    pp: np.ndarray  # complex128[n_series][n_knots-1][4]
    desc: np.ndarray  # structured [batch][N]: drive/det/off series + scales
    interaction: np.ndarray  # float64[n_mats][N][N]
    dissipator: np.ndarray | None  # complex128[4][4] or None
    series_knots: list[np.ndarray]  # the knot arrays (for tests / bounds)
    collapse_local: np.ndarray | None = None  # complex128[n_ops][2][2] (Monte-Carlo solver)
    dterms: np.ndarray | None = None  # DTERM_DTYPE[n]: extra detuning terms (hf detuning noise)


DESC_DTYPE = np.dtype(
    [
        ("drive_series", "<i4"),
        ("det_series", "<i4"),
        ("off_series", "<i4"),
        ("extra", "<i4"),  # 1-based index of the first extra detuning term (0 = none)
        ("drive_scale", "<f8"),
        ("det_scale", "<f8"),
        ("off_scale", "<f8"),
    ],
    align=True,
)


# extra detuning terms (include/rydemu.h: ryd_dterm)
DTERM_DTYPE = np.dtype([("series", "<i4"), ("remaining", "<i4"), ("scale", "<f8")], align=True)


class _SeriesPool:
    """Deduplicated pool of complex knot arrays."""

    def __init__(self) -> None:
        self.arrays: list[np.ndarray] = []
        self._index: dict[bytes, int] = {}

    def add(self, knots: np.ndarray) -> int:
        arr = np.ascontiguousarray(knots, dtype=np.complex128)
        if not np.any(arr != 0):
            return -1  # all-zero coefficient: term dropped (hamiltonian.py:354, 377)
        key = arr.tobytes()
        if key not in self._index:
            self._index[key] = len(self.arrays)
            self.arrays.append(arr)
        return self._index[key]


def local_collapse_ops(
    collapse_ops: Sequence[tuple[Any, Any]],
    eigenbasis: Sequence[str],
    paulis: Mapping[str, Sequence[tuple[complex, str]]] | None = None,
) -> np.ndarray | None:
    """The local d x d collapse operators ``coeff * op`` themselves,
    complex128[n_ops][d][d], expanded as ``Hamiltonian._build_collapse_operators``
    does (hamiltonian.py:97-124) - what ``qutip.mcsolve`` receives as ``c_ops``,
    one copy per atom."""
    if not collapse_ops:
        return None
    d = len(eigenbasis)
    proj: dict[str, np.ndarray] = {}
    for i, a in enumerate(eigenbasis):
        for j, b in enumerate(eigenbasis):
            m = np.zeros((d, d), dtype=complex)
            m[i, j] = 1.0
            proj["sigma_" + a + b] = m
    out = []
    for coeff, op in collapse_ops:
        if isinstance(op, str):
            if op in proj:
                c = coeff * proj[op]
            else:
                c = sum(coeff * pc * proj[po] for pc, po in (paulis or {})[op])
        else:
            c = coeff * np.asarray(op, dtype=complex)
        out.append(np.asarray(c, dtype=np.complex128))
    return np.ascontiguousarray(np.stack(out))


def local_dissipator(
    collapse_ops: Sequence[tuple[Any, Any]],
    eigenbasis: Sequence[str],
    paulis: Mapping[str, Sequence[tuple[complex, str]]] | None = None,
) -> np.ndarray | None:
    """4x4 local Lindblad superoperator for d = 2 from the (coeff, op) specs of
    ``HamiltonianData._build_local_collapse_operators``
    (pulser-core/pulser/_hamiltonian_data/hamiltonian_data.py:654-739), expanded
    as ``Hamiltonian._build_collapse_operators`` does (hamiltonian.py:97-124).
    Index = 2*a + b with a (row) and b (column) the local basis indices."""
    if not collapse_ops:
        return None
    d = len(eigenbasis)
    if d != 2:
        raise NotImplementedError(
            "Only 2-level bases are supported by the MI355X backend so far "
            f"(got eigenbasis {list(eigenbasis)})."
        )
    proj: dict[str, np.ndarray] = {}
    for i, a in enumerate(eigenbasis):
        for j, b in enumerate(eigenbasis):
            m = np.zeros((d, d), dtype=complex)
            m[i, j] = 1.0
            proj["sigma_" + a + b] = m
    S = np.zeros((4, 4), dtype=complex)
    eye = np.eye(2)
    for coeff, op in collapse_ops:
        if isinstance(op, str):
            if op in proj:
                c = coeff * proj[op]
            else:
                c = sum(coeff * pc * proj[po] for pc, po in (paulis or {})[op])
        else:
            c = coeff * np.asarray(op, dtype=complex)
        cdc = c.conj().T @ c
        # rho'[a,b] = sum C[a,a'] rho[a',b'] conj(C[b,b']) - 1/2 (CdC rho + rho CdC)
        S += np.kron(c, c.conj()) - 0.5 * np.kron(cdc, eye) - 0.5 * np.kron(eye, cdc.T)
    return S


def lower(problems: Sequence[Mapping[str, Any]]) -> DeviceTables:
    """Lower a batch of problems (same register size, duration and sampling
    rate; e.g. the noise trajectories of one sequence) to device tables."""
    if not problems:
        raise ValueError("At least one problem is required.")
    p0 = problems[0]
    n = int(p0["n_qudits"])
    duration = int(p0["duration"])
    rate = float(p0.get("sampling_rate", 1.0))
    basis_name = p0["basis_name"]
    if basis_name not in SUPPORTED_BASES or len(p0["eigenbasis"]) != 2:
        raise NotImplementedError(
            f"Basis '{basis_name}' (eigenbasis {list(p0['eigenbasis'])}) is not "
            "supported by the MI355X backend yet; supported: "
            f"{SUPPORTED_BASES} without leakage."
        )
    tknots = sampling_times(duration, rate)
    pool = _SeriesPool()
    desc = np.zeros((len(problems), n), dtype=DESC_DTYPE)
    desc["drive_series"] = desc["det_series"] = desc["off_series"] = -1
    mats = []
    for b, p in enumerate(problems):
        if (
            int(p["n_qudits"]) != n
            or int(p["duration"]) != duration
            or float(p.get("sampling_rate", 1.0)) != rate
            or p["basis_name"] != basis_name
        ):
            raise ValueError("All problems of a batch must share N, duration, "
                             "sampling rate and basis.")
        if int(p.get("slm_end", 0)) > 0 and p.get("interaction_type") == "XY":
            raise NotImplementedError("XY mode is not supported yet.")
        samples = p["samples"]
        per_qubit: list[list[tuple[np.ndarray, np.ndarray]]] = [[] for _ in range(n)]
        for addr in samples:
            for basis, s in samples[addr].items():
                if not s:
                    continue
                if basis != basis_name:
                    raise NotImplementedError(
                        f"Samples address basis '{basis}' but the state basis "
                        f"is '{basis_name}'."
                    )
                if addr == "Global":
                    c = 0.5 * np.asarray(s["amp"]) * np.exp(-1j * np.asarray(s["phase"]))
                    dt = np.asarray(s["det"], dtype=float)
                    for k in range(n):
                        per_qubit[k].append((c, dt))
                else:
                    for q, sq in s.items():
                        c = 0.5 * np.asarray(sq["amp"]) * np.exp(-1j * np.asarray(sq["phase"]))
                        per_qubit[int(q)].append((c, np.asarray(sq["det"], dtype=float)))
        for k in range(n):
            if not per_qubit[k]:
                continue
            # several channels on one atom add up (QobjEvo sums the terms)
            c = sum(x[0] for x in per_qubit[k])
            dt = sum(x[1] for x in per_qubit[k])
            di = pool.add(adapt_to_sampling_rate(c, rate, duration))
            ti = pool.add(adapt_to_sampling_rate(dt, rate, duration))
            desc[b, k]["drive_series"] = di
            desc[b, k]["drive_scale"] = 1.0 if di >= 0 else 0.0
            desc[b, k]["det_series"] = ti
            desc[b, k]["det_scale"] = 1.0 if ti >= 0 else 0.0
        bad = np.asarray(p.get("bad_atoms", np.zeros(n, bool)), dtype=bool)
        u = np.array(p["interaction_matrix"], dtype=float)[-1].copy()
        np.fill_diagonal(u, 0.0)
        if "digital" in basis_name or (n - int(bad.sum())) <= 1:
            u[:] = 0.0  # hamiltonian.py:393-396
        u[bad, :] = 0.0
        u[:, bad] = 0.0  # hamiltonian.py:313-325
        mats.append(u)
    if not pool.arrays:  # no drive at all: H = interaction only
        pool.arrays.append(np.zeros(len(tknots), dtype=np.complex128))
    pp = np.empty((len(pool.arrays), len(tknots) - 1, 4), dtype=np.complex128)
    for i, knots in enumerate(pool.arrays):
        cs = CubicSpline(tknots, knots, bc_type="not-a-knot")
        pp[i] = np.transpose(cs.c, (1, 0))
    shared = all(np.array_equal(mats[0], m) for m in mats[1:])
    interaction = np.stack(mats[:1] if shared else mats)
    S = local_dissipator(
        p0.get("collapse_ops", []), p0["eigenbasis"], p0.get("depolarizing_pauli_2ds")
    )
    return DeviceTables(
        n_qubits=n,
        batch=len(problems),
        tknots=np.ascontiguousarray(tknots, dtype=np.float64),
        pp=np.ascontiguousarray(pp),
        desc=desc,
        interaction=np.ascontiguousarray(interaction),
        dissipator=S,
        series_knots=pool.arrays,
        collapse_local=local_collapse_ops(
            p0.get("collapse_ops", []), p0["eigenbasis"], p0.get("depolarizing_pauli_2ds")
        ),
    )
