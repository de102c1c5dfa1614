"""``QutipBackendV2``-style emulator backend on the MI355X engine.

Mirrors the non-deprecated entry point of the reference,
``pulser_simulation.QutipBackendV2`` (pulser-simulation/pulser_simulation/
qutip_backend.py:121-325) with its config (``qutip_config.py:28-192``), state
type (``qutip_state.py``), the default observables
(pulser-core/pulser/backend/default_observables.py) and ``Results``
(pulser-core/pulser/backend/results.py): per evaluation time the normalised
state (``.unit()``, qutip_backend.py:257) and the *noiseless* Hamiltonian
operator (``:259-264``) are handed to every observable; noisy trajectories
produce one ``Results`` each and are aggregated (MEAN for numbers, BAG_UNION
for Counters, mean of |psi><psi| for the state, ``:322-325``).

The solver call is the HIP engine; observables that need ``H|psi>`` apply the
matrix-free generator kernel on the device (``ryd_apply_generator``).
"""

from __future__ import annotations

import collections.abc
import enum
import json
import math
import uuid
import warnings
from collections import Counter, defaultdict
from typing import Any, Callable, Mapping, Sequence

import numpy as np

from .noise_model import NoiseModel, has_stochastic_noise
from .results import DeviceState, QState, multinomial
from .simulation import QutipEmulator, Solver

__all__ = [
    "QutipBackendV2", "QutipBackend", "EmulatorConfig", "QutipConfig", "RydState", "RydOperator", "Results", "Observable", "StateResult",
    "AggregationMethod", "BitStrings", "Fidelity", "Expectation", "CorrelationMatrix", "Occupation", "Energy",
    "EnergyVariance", "EnergySecondMoment",
]

_ONE_STATE = {("r", "g"): "r", ("g", "h"): "h", ("u", "d"): "d"}


def _validate_eigenstates(eigenstates: Any) -> None:
    """pulser/backend/state.py:196-208."""
    if not isinstance(eigenstates, collections.abc.Sequence):
        raise TypeError(
            "'eigenstates' must be a 'collections.Sequence' "
            f"(list or tuple), not {type(eigenstates).__name__}."
        )
    if any(not isinstance(s, str) or len(s) != 1 for s in eigenstates):
        raise ValueError("All eigenstates must be represented by single characters.")
    if len(eigenstates) != len(set(eigenstates)):
        raise ValueError("'eigenstates' can't contain repeated entries.")


def _validate_shape(shape: tuple[int, ...], qudit_dim: int) -> None:
    """qutip_state.py:272-280."""
    n = math.log(shape[0], qudit_dim)
    if not np.isclose(n, round(n)):
        raise ValueError(
            f"An array with shape {tuple(shape)} is incompatible with "
            f"a system of {qudit_dim}-level qudits."
        )


# ------------------------------------------------------------------- state
class RydState:
    """``QutipState`` (pulser_simulation/qutip_state.py:38-281) on a NumPy state."""

    def __init__(self, state: Any, *, eigenstates: Sequence[str]) -> None:
        _validate_eigenstates(eigenstates)
        self.eigenstates = tuple(eigenstates)
        if isinstance(state, DeviceState):  # a density matrix that lives on the GPU (aggregated results)
            self._state = state
            _validate_shape(state.shape, len(self.eigenstates))
            self._n = int(round(math.log(state.shape[0], len(self.eigenstates))))
            self._amplitudes = None
            return
        arr = np.asarray(state)
        if isinstance(state, (str, bytes)) or arr.dtype == object or arr.ndim not in (1, 2):
            raise TypeError(
                "'state' must be a state vector (ket or bra) or a density matrix, "
                f"not {state!r}."
            )
        if arr.ndim == 2 and arr.shape[0] == 1 and arr.shape[1] > 1:  # a bra
            arr = arr.conj().T
        if arr.ndim == 2 and arr.shape[1] not in (1, arr.shape[0]):
            raise TypeError(
                "'state' must be a state vector (ket or bra) or a density matrix, "
                f"not an array of shape {arr.shape}."
            )
        self._state = QState(arr)
        _validate_shape(self._state.shape, len(self.eigenstates))
        self._n = int(round(math.log(self._state.shape[0], len(self.eigenstates))))
        self._amplitudes: Mapping[str, complex] | None = None

    def _to_abstract_repr(self) -> dict[str, Any]:
        """pulser/backend/state.py:234-254: only states built by
        ``from_state_amplitudes`` (and not modified since) can be serialised."""
        if self._amplitudes is None:
            raise ValueError(
                "Failed to serialize state of type 'RydState' because it was not created "
                "via 'RydState.from_state_amplitudes()'.")
        stashed = self.from_state_amplitudes(eigenstates=self.eigenstates,
                                             amplitudes=self._amplitudes)
        if abs(float(self.overlap(stashed)) - 1.0) > 1e-12:
            raise ValueError(
                "Failed to serialize state of type 'RydState' because it was modified in "
                "place after its creation.")
        return {"eigenstates": tuple(self.eigenstates), "amplitudes": dict(self._amplitudes)}

    @property
    def n_qudits(self) -> int:
        return self._n

    @property
    def qudit_dim(self) -> int:
        return len(self.eigenstates)

    def to_qobj(self) -> QState:
        return self._state

    def infer_one_state(self) -> str:
        """pulser/backend/state.py: the eigenstate measured as 1."""
        eig = set(self.eigenstates) - {"x"}  # the leakage state never counts
        if eig == {"0", "1"}:
            return "1"
        for pair, one in _ONE_STATE.items():
            if set(pair) == eig:
                return one
        raise RuntimeError(f"Failed to infer the 'one state' from the eigenstates: {self.eigenstates}")

    def get_basis_state_from_index(self, index: int) -> str:
        if index < 0:
            raise ValueError(f"'index' must be a non-negative integer; got {index} instead.")
        d = self.qudit_dim
        digits = np.base_repr(index, base=d).zfill(self._n)
        return "".join(self.eigenstates[int(c)] for c in digits)

    def overlap(self, other: "RydState") -> float:
        """qutip_state.py:86-110."""
        if not isinstance(other, RydState):
            raise TypeError(f"'RydState.overlap()' expects another 'RydState', not {type(other)}.")
        if self.n_qudits != other.n_qudits or self.qudit_dim != other.qudit_dim:
            raise ValueError(
                "Can't calculate the overlap between a state with "
                f"{self.n_qudits} {self.qudit_dim}-dimensional qudits and "
                f"another with {other.n_qudits} {other.qudit_dim}-dimensional qudits."
            )
        if self.eigenstates != other.eigenstates:
            msg = ("Can't calculate the overlap between states with eigenstates "
                   f"{self.eigenstates} and {other.eigenstates}.")
            if set(self.eigenstates) != set(other.eigenstates):
                raise ValueError(msg)
            raise NotImplementedError(msg)
        ov = self._state.overlap(other._state)
        if self._state.isket and other._state.isket:
            ov = np.abs(ov) ** 2
        return float(np.real(ov))

    def probabilities(self, *, cutoff: float = 1e-12) -> dict[str, float]:
        """qutip_state.py:112-141."""
        if not self._state.isket:
            probs = np.abs(self._state.diag()).real
        else:
            probs = (np.abs(self._state.full()) ** 2).flatten().real
        non_zero = np.argwhere(probs > cutoff).flatten()
        probs = probs[non_zero]
        probs = probs / np.sum(probs)
        return dict(zip(map(self.get_basis_state_from_index, non_zero), probs))

    def bitstring_probabilities(self, *, one_state: str | None = None,
                                cutoff: float = 1e-12) -> Mapping[str, float]:
        """qutip_state.py:143-167."""
        one_state = one_state or self.infer_one_state()
        zero_states = set(self.eigenstates) - {one_state}
        probs = self.probabilities(cutoff=cutoff)
        out: dict[str, float] = defaultdict(float)
        for state_str in probs:
            bitstring = state_str.replace(one_state, "1")
            for s_ in zero_states:
                bitstring = bitstring.replace(s_, "0")
            out[bitstring] += probs[state_str]
        return dict(out)

    def sample(self, *, num_shots: int, one_state: str | None = None,
               p_false_pos: float = 0.0, p_false_neg: float = 0.0) -> Counter:
        """qutip_state.py:169-217 (same NumPy call sequence)."""
        bitstring_probs = self.bitstring_probabilities(one_state=one_state,
                                                       cutoff=1 / (1000 * num_shots))
        bitstrings = np.array(list(bitstring_probs))
        probs = np.array(list(map(float, bitstring_probs.values())))
        indices = multinomial(num_shots, probs)
        if p_false_pos == 0.0 and p_false_neg == 0.0:
            return Counter(bitstrings[indices].tolist())
        bitstr_arr = np.array([list(bs) for bs in bitstrings[indices]], dtype=int)
        flip_probs = np.where(bitstr_arr == 1, p_false_neg, p_false_pos)
        random_matrix = np.random.uniform(size=flip_probs.shape)
        new_bitstrings = bitstr_arr ^ (random_matrix < flip_probs)
        new_counts: Counter = Counter(map(tuple, new_bitstrings))
        return Counter({"".join(map(str, k)): v for k, v in new_counts.items()})

    @classmethod
    def from_state_amplitudes(cls, *, eigenstates: Sequence[str],
                              amplitudes: Mapping[str, complex]) -> "RydState":
        """pulser/backend/state.py:143-176, 217-232: {"rgr": a, ...} -> ket with exactly
        these amplitudes (no normalisation, like the reference)."""
        _validate_eigenstates(eigenstates)
        keys = list(amplitudes)
        n = len(keys[0])
        if not all(len(k) == n and set(k) <= set(eigenstates) for k in keys):
            raise ValueError(
                "All basis states must be combinations of eigenstates with the"
                f" same length. Expected combinations of {eigenstates}, each "
                f"with {n} elements."
            )
        d = len(eigenstates)
        vec = np.zeros(d**n, dtype=complex)
        amps = {k: complex(v) for k, v in amplitudes.items()}
        for key, amp in amps.items():
            idx = 0
            for ch in key:
                idx = idx * d + list(eigenstates).index(ch)
            vec[idx] += amp
        out = cls(QState(vec), eigenstates=eigenstates)
        out._amplitudes = amps
        return out

    def __repr__(self) -> str:
        return "\n".join(["RydState", "--------", f"Eigenstates: {self.eigenstates}",
                          repr(np.asarray(self._state))])

    def __eq__(self, other: Any) -> bool:
        if not isinstance(other, RydState):
            return False
        return (self.eigenstates == other.eigenstates and self._state.shape == other._state.shape
                and np.allclose(np.asarray(self._state), np.asarray(other._state), rtol=0, atol=1e-12))

    __hash__ = None  # type: ignore[assignment]


class RydOperator:
    """``QutipOperator`` (pulser_simulation/qutip_op.py:30-260) on a SciPy sparse
    matrix: the operator type of ``Expectation`` and of custom callbacks."""

    def __init__(self, operator: Any, *, eigenstates: Sequence[str]) -> None:
        import scipy.sparse as sp

        _validate_eigenstates(eigenstates)
        self.eigenstates = tuple(eigenstates)
        if isinstance(operator, (str, bytes)) or (not sp.issparse(operator) and np.asarray(operator).ndim != 2):
            raise TypeError(f"'operator' must be a square matrix (dense or sparse), not {operator!r}.")
        self._operator = sp.csr_matrix(operator, dtype=complex)
        if self._operator.shape[0] != self._operator.shape[1]:
            raise TypeError(f"'operator' must be a square matrix, not one of shape {self._operator.shape}.")
        _validate_shape(self._operator.shape, len(self.eigenstates))
        self._n_qudits: int | None = None
        self._operations: Any = None

    def to_qobj(self) -> Any:
        return self._operator

    def _validate_other(self, other: Any, expected: type, op_name: str) -> None:
        """qutip_op.py:240-260."""
        if not isinstance(other, expected):
            raise TypeError(f"'{op_name}' expects a '{expected.__name__}' instance, not {type(other)}.")
        if self.eigenstates != other.eigenstates:
            msg = (f"Can't apply {op_name} between a {self.__class__.__name__} with eigenstates "
                   f"{self.eigenstates} and a {other.__class__.__name__} with {other.eigenstates}.")
            if set(self.eigenstates) != set(other.eigenstates):
                raise ValueError(msg)
            raise NotImplementedError(msg)

    def apply_to(self, state: RydState) -> RydState:
        """O|psi> or O rho O^dag (qutip_op.py:75-88)."""
        self._validate_other(state, RydState, "RydOperator.apply_to()")
        a = np.asarray(state.to_qobj())
        out = self._operator @ a
        if not state.to_qobj().isket:
            out = (self._operator @ out.conj().T).conj().T
        return RydState(out, eigenstates=state.eigenstates)

    def expect(self, state: RydState) -> complex:
        """<psi|O|psi> or Tr(O rho) (qutip_op.py:90-100)."""
        self._validate_other(state, RydState, "RydOperator.expect()")
        a = np.asarray(state.to_qobj())
        if state.to_qobj().isket:
            return complex(np.vdot(a, self._operator @ a))
        return complex((self._operator @ a).trace())

    def __add__(self, other: "RydOperator") -> "RydOperator":
        self._validate_other(other, RydOperator, "__add__")
        return RydOperator(self._operator + other._operator, eigenstates=self.eigenstates)

    def __rmul__(self, scalar: complex) -> "RydOperator":
        return RydOperator(complex(scalar) * self._operator, eigenstates=self.eigenstates)

    def __matmul__(self, other: "RydOperator") -> "RydOperator":
        self._validate_other(other, RydOperator, "__matmul__")
        return RydOperator(self._operator @ other._operator, eigenstates=self.eigenstates)

    def __eq__(self, other: Any) -> bool:
        return (isinstance(other, RydOperator) and self.eigenstates == other.eigenstates
                and (self._operator != other._operator).nnz == 0)

    __hash__ = None  # type: ignore[assignment]

    @classmethod
    def from_operator_repr(cls, *, eigenstates: Sequence[str], n_qudits: int,
                           operations: Sequence[tuple[complex, Sequence[tuple[Mapping[str, complex], Any]]]]
                           ) -> "RydOperator":
        """pulser/backend/operator.py:115-186 + qutip_op.py:149-220:
        ``operations = [(coeff, [({"rr": 1.0, ...}, {qudit indices}), ...]), ...]`` - a
        weighted sum of tensor products of single-qudit operators ``|i><j|``."""
        import scipy.sparse as sp

        eigenstates = tuple(eigenstates)
        d = len(eigenstates)
        for num, (_, tensor_op) in enumerate(operations):  # operator.py:205-235
            free = set(range(n_qudits))
            for qudit_op, inds in tensor_op:
                if bad := (set(inds) - free):
                    raise ValueError(
                        f"Got invalid indices for a system with {n_qudits} qudits: {bad}. For "
                        f"TensorOp #{num}, only indices {free} were still available.")
                free.difference_update(inds)
                for proj in qudit_op:
                    if len(proj) != 2 or any(c not in eigenstates for c in proj):
                        raise ValueError(
                            "Every QuditOp key must be made up of two eigenstates among "
                            f"{eigenstates}; instead, got '{proj}'.")
        full = sp.csr_matrix((d**n_qudits, d**n_qudits), dtype=complex)
        eye = sp.identity(d, dtype=complex, format="csr")
        rebuilt = []
        for coeff, tensor_op in operations:
            factors = [eye] * n_qudits
            re_tensor = []
            for qudit_op, inds in tensor_op:
                local = np.zeros((d, d), dtype=complex)
                for proj, c in qudit_op.items():
                    local[eigenstates.index(proj[0]), eigenstates.index(proj[1])] += complex(c)
                for i in inds:
                    factors[i] = sp.csr_matrix(local)
                re_tensor.append(({k: complex(v) for k, v in qudit_op.items()}, set(inds)))
            term = sp.identity(1, dtype=complex, format="csr")
            for f in factors:
                term = sp.kron(term, f, format="csr")
            full = full + complex(coeff) * term
            rebuilt.append((complex(coeff), re_tensor))
        out = cls(full, eigenstates=eigenstates)
        out._n_qudits, out._operations = int(n_qudits), rebuilt
        return out

    def _to_abstract_repr(self) -> dict[str, Any]:
        """operator.py:188-203."""
        if self._operations is None:
            raise ValueError(
                "Failed to serialize state of type 'RydOperator' because it was not created "
                "via 'RydOperator.from_operator_repr()'.")
        return {"eigenstates": tuple(self.eigenstates), "n_qudits": self._n_qudits,
                "operations": self._operations}

    def __repr__(self) -> str:
        return "\n".join(["RydOperator", "-----------", f"Eigenstates: {self.eigenstates}",
                          repr(self._operator)])


class HamiltonianOperator:
    """The noiseless H(t) handed to observables (qutip_backend.py:259-264),
    matrix-free: ``apply_to`` runs the device generator kernel."""

    def __init__(self, engine: Any, t_us: float, eigenstates: Sequence[str]) -> None:
        self._eng, self._t, self.eigenstates = engine, t_us, tuple(eigenstates)
        self._observed: tuple[int, dict[str, Any]] | None = None

    def observe(self, state: RydState) -> dict[str, Any] | None:
        """Device-side values behind Occupation / CorrelationMatrix / Energy* for a ket of a
        2-level register: ONE ``ryd_observe`` call per (state, evaluation time), shared by every
        observable asking for it (kets: pair reduction + one generator application + one dot;
        density matrices: pair reduction on the diagonal + one gather kernel for Tr(H rho), Tr(H^2 rho)).
        None for multi-level registers: the host formulas are used then."""
        if not hasattr(self._eng, "observe") or len(self.eigenstates) != 2:
            return None
        if self._observed is not None and self._observed[0] == id(state):
            return self._observed[1]
        q = state.to_qobj()
        if q.shape[0] != self._eng.dim:
            return None
        import torch

        if q.isket:
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(q)[:, 0][None, :])).to(self._eng.device)
            raw = self._eng.observe(x, self._t)
        else:  # density matrix of a master-equation run, observed with the noiseless ket engine's H(t)
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(q)[None, :, :])).to(self._eng.device)
            raw = self._eng.observe(x, self._t, density=True)
        n2 = float(raw["norm2"][0])
        res = {"occupation": raw["occupation"][0] / n2, "correlation": raw["correlation"][0] / n2,
               "energy": float(raw["energy"][0]) / n2, "energy2": float(raw["energy2"][0]) / n2}
        self._observed = (id(state), res)
        return res

    def _h_on(self, arr: np.ndarray) -> np.ndarray:
        """H @ arr for a ket (D,1) or a matrix (D,D) of column vectors."""
        import torch

        cols = np.ascontiguousarray(np.asarray(arr, dtype=complex).T)  # rows = kets
        out = np.empty_like(cols)
        for i in range(cols.shape[0]):
            x = torch.from_numpy(cols[i:i + 1].copy()).to(self._eng.device)
            out[i] = 1j * self._eng.apply_generator(x, self._t).cpu().numpy()[0]  # G = -iH
        return out.T

    def apply_to(self, state: RydState) -> RydState:
        s = state.to_qobj()
        if s.isket:
            return RydState(self._h_on(s), eigenstates=state.eigenstates)
        h_rho = self._h_on(np.asarray(s))
        return RydState(self._h_on(h_rho.conj().T).conj().T, eigenstates=state.eigenstates)  # H rho H

    def expect(self, state: RydState) -> float:
        s = state.to_qobj()
        if s.isket:
            return float(np.real(np.vdot(np.asarray(s), self._h_on(s))))
        return float(np.real(np.trace(self._h_on(np.asarray(s)))))


# --------------------------------------------------------------- observables
TIME_TOLERANCE = 1e-12  # observable.py:33


def _validate_eval_times(evaluation_times: Any) -> np.ndarray:
    """observable.py:219-242: relative times in [0, 1], distinct up to 1e-12, ascending."""
    ev = np.array(evaluation_times, dtype=float)
    if np.any((ev < 0.0) | (ev > 1.0)):
        raise ValueError(f"All evaluation times must be between 0. and 1. Instead, got {evaluation_times!r}.")
    if np.any(np.abs(ev[:-1] - ev[1:]) < TIME_TOLERANCE):
        raise ValueError(f"Evaluation times must be unique up to {TIME_TOLERANCE} but "
                         f"{evaluation_times!r} has repeated values.")
    if not np.all(ev[:-1] < ev[1:]):
        raise ValueError(f"Evaluation times must be in ascending order.Instead, got {evaluation_times!r}.")
    return ev


class Observable:
    """pulser/backend/observable.py:60-215."""

    default_aggregation = "mean"

    def __init__(self, *, evaluation_times: Sequence[float] | None = None,
                 tag_suffix: str | None = None, default_aggregation_method: Any = None) -> None:
        if evaluation_times is not None:
            self.evaluation_times: np.ndarray | None = _validate_eval_times(evaluation_times)
        else:
            self.evaluation_times = None
        self._tag_suffix = tag_suffix
        self._uuid = uuid.uuid4()
        if default_aggregation_method is not None:  # per-instance override (observable.py:100-131)
            kind = default_aggregation_method
            self.default_aggregation = kind if isinstance(kind, str) else _AGG_KIND[int(kind)]

    @property
    def default_aggregation_method(self) -> "AggregationMethod":
        """How results of this observable are combined over trajectories (read-only)."""
        return AggregationMethod(_AGG_CODE[self.default_aggregation])

    _base_tag = "observable"

    def _to_abstract_repr(self) -> dict[str, Any]:
        """pulser/backend/observable.py:132-139."""
        return {
            "observable": self._base_tag,
            "evaluation_times": None if self.evaluation_times is None else self.evaluation_times.tolist(),
            "tag_suffix": self._tag_suffix,
            "default_aggregation_method": _AGG_CODE[self.default_aggregation],
            "uuid": str(self._uuid),
        }

    @property
    def tag(self) -> str:
        return self._base_tag if self._tag_suffix is None else f"{self._base_tag}_{self._tag_suffix}"

    @property
    def uuid(self) -> uuid.UUID:
        return self._uuid

    def __repr__(self) -> str:
        return f"{self.tag}:{self._uuid}"

    def __call__(self, config: "QutipConfig", t: float, state: RydState,
                 hamiltonian: HamiltonianOperator, result: "Results") -> None:
        if self._fires(config, t, result.total_duration):
            result._store(observable=self, time=t,
                          value=self.apply(config=config, state=state, hamiltonian=hamiltonian))

    def _fires(self, config: "QutipConfig", t: float, total_duration: int) -> bool:
        """observable.py:178-193: is relative time ``t`` one of this observable's evaluation times."""
        time_tol = (0.5 / total_duration) if total_duration else 1e-6
        if self.evaluation_times is not None:
            return bool(config.is_time_in_evaluation_times(t, self.evaluation_times, tol=time_tol))
        return bool(config.is_evaluation_time(t, tol=time_tol))

    def apply(self, *, config: "QutipConfig", state: RydState,
              hamiltonian: HamiltonianOperator) -> Any:  # pragma: no cover
        raise NotImplementedError


class StateResult(Observable):
    _base_tag = "state"
    default_aggregation = "density_matrix"

    def _to_abstract_repr(self) -> dict[str, Any]:
        raise ValueError(  # default_observables.py:66-73
            "`StateResult` observable is not supported in any remote backend. If you are "
            "interested in the full quantum state at arbitrary times during the emulation, "
            "please, consider using the local version of the same backend.")

    def apply(self, *, state: RydState, **kw: Any) -> RydState:
        return RydState(np.array(state.to_qobj()), eigenstates=state.eigenstates)


class BitStrings(Observable):
    _base_tag = "bitstrings"
    default_aggregation = "bag_union"

    def __init__(self, *, evaluation_times: Sequence[float] | None = None,
                 num_shots: int | None = None, one_state: str | None = None,
                 tag_suffix: str | None = None,
                 default_aggregation_method: Any = None) -> None:
        super().__init__(evaluation_times=evaluation_times, tag_suffix=tag_suffix,
                         default_aggregation_method=default_aggregation_method)
        if num_shots is not None and num_shots < 1:
            raise ValueError(f"'num_shots' must be greater than or equal to 1, not {num_shots}.")
        self._num_shots = None if num_shots is None else int(num_shots)
        self.one_state = one_state

    def _to_abstract_repr(self) -> dict[str, Any]:
        d = super()._to_abstract_repr()
        d["num_shots"] = self._num_shots
        d["one_state"] = self.one_state
        return d

    def _skip_draws(self, config: "QutipConfig", n_qudits: int) -> None:
        """Advance the global ``np.random`` stream exactly as :meth:`apply` would, without a state
        (``RydState.sample``: ``rand(num_shots)``, then ``uniform(size=(num_shots, N))`` when there
        are measurement errors) - used by ranks that do not own a trajectory of a sharded run."""
        shots = self._num_shots if self._num_shots is not None else config.default_num_shots
        np.random.rand(shots)
        if not (config.noise_model.p_false_pos == 0.0 and config.noise_model.p_false_neg == 0.0):
            np.random.uniform(size=(shots, n_qudits))

    def apply(self, *, config: "QutipConfig", state: RydState, **kw: Any) -> Counter:
        return state.sample(
            num_shots=self._num_shots if self._num_shots is not None else config.default_num_shots,
            one_state=self.one_state,
            p_false_pos=config.noise_model.p_false_pos,
            p_false_neg=config.noise_model.p_false_neg,
        )


class Fidelity(Observable):
    _base_tag = "fidelity"

    def __init__(self, state: RydState, *, evaluation_times: Sequence[float] | None = None,
                 tag_suffix: str | None = None,
                 default_aggregation_method: Any = None) -> None:
        super().__init__(evaluation_times=evaluation_times, tag_suffix=tag_suffix,
                         default_aggregation_method=default_aggregation_method)
        if not isinstance(state, RydState):
            raise TypeError(f"'state' must be a State instance; got {type(state)} instead.")
        self.state = state

    def _to_abstract_repr(self) -> dict[str, Any]:
        d = super()._to_abstract_repr()
        d["state"] = self.state
        return d

    def apply(self, *, state: RydState, **kw: Any) -> float:
        return self.state.overlap(state)


class Expectation(Observable):
    """<O> of an operator (default_observables.py:239-288): a :class:`RydOperator`
    (e.g. from ``from_operator_repr``) or, as a convenience, a dense d^N x d^N array."""

    _base_tag = "expectation"

    def __init__(self, operator: Any, *, evaluation_times: Sequence[float] | None = None,
                 tag_suffix: str | None = None,
                 default_aggregation_method: Any = None) -> None:
        super().__init__(evaluation_times=evaluation_times, tag_suffix=tag_suffix,
                         default_aggregation_method=default_aggregation_method)
        if not isinstance(operator, (RydOperator, np.ndarray)):
            raise TypeError(f"'operator' must be an Operator instance; got {type(operator)} instead.")
        self.operator = operator

    def _to_abstract_repr(self) -> dict[str, Any]:
        if not isinstance(self.operator, RydOperator):
            raise ValueError("An 'Expectation' of a dense matrix has no abstract representation; "
                             "build the operator with 'RydOperator.from_operator_repr()'.")
        d = super()._to_abstract_repr()
        d["operator"] = self.operator
        return d

    def apply(self, *, state: RydState, **kw: Any) -> Any:
        if isinstance(self.operator, RydOperator):
            return self.operator.expect(state)
        s = np.asarray(state.to_qobj())
        op = np.asarray(self.operator, dtype=complex)
        if s.shape[1] == 1:
            return complex(np.vdot(s, op @ s))
        return complex(np.trace(op @ s))


def _probabilities(state: RydState) -> np.ndarray:
    s = state.to_qobj()
    return (np.abs(np.asarray(s)[:, 0]) ** 2) if s.isket else np.real(s.diag())


def _one_mask(state: RydState, one_state: str | None) -> np.ndarray:
    """bool[D, N]: atom k of basis state i is in the 'one' eigenstate."""
    one = one_state or state.infer_one_state()
    d, n = state.qudit_dim, state.n_qudits
    idx = np.arange(d**n)
    digits = np.stack([(idx // d ** (n - 1 - k)) % d for k in range(n)], axis=1)
    return digits == list(state.eigenstates).index(one)


class Occupation(Observable):
    """<n_i> with n = |one><one| (default_observables.py:377-435)."""

    _base_tag = "occupation"

    def __init__(self, *, evaluation_times: Sequence[float] | None = None,
                 one_state: str | None = None, tag_suffix: str | None = None,
                 default_aggregation_method: Any = None) -> None:
        super().__init__(evaluation_times=evaluation_times, tag_suffix=tag_suffix,
                         default_aggregation_method=default_aggregation_method)
        self.one_state = one_state

    def _to_abstract_repr(self) -> dict[str, Any]:
        d = super()._to_abstract_repr()
        d["one_state"] = self.one_state
        return d

    def apply(self, *, state: RydState, hamiltonian: Any = None, **kw: Any) -> list:
        dev = hamiltonian.observe(state) if hasattr(hamiltonian, "observe") else None
        if dev is not None:  # device reduction; local state 0 is what the kernel counts
            one = self.one_state or state.infer_one_state()
            occ0 = dev["occupation"]
            return [float(v) for v in (occ0 if list(state.eigenstates).index(one) == 0 else 1.0 - occ0)]
        p = _probabilities(state)
        return [float(v) for v in p @ _one_mask(state, self.one_state)]


class CorrelationMatrix(Observable):
    """<n_i n_j> (default_observables.py:291-374)."""

    _base_tag = "correlation_matrix"

    def __init__(self, *, evaluation_times: Sequence[float] | None = None,
                 one_state: str | None = None, tag_suffix: str | None = None,
                 default_aggregation_method: Any = None) -> None:
        super().__init__(evaluation_times=evaluation_times, tag_suffix=tag_suffix,
                         default_aggregation_method=default_aggregation_method)
        self.one_state = one_state

    def _to_abstract_repr(self) -> dict[str, Any]:
        d = super()._to_abstract_repr()
        d["one_state"] = self.one_state
        return d

    def apply(self, *, state: RydState, hamiltonian: Any = None, **kw: Any) -> list[list]:
        dev = hamiltonian.observe(state) if hasattr(hamiltonian, "observe") else None
        if dev is not None:
            one = self.one_state or state.infer_one_state()
            c0, o0 = dev["correlation"], dev["occupation"]
            if list(state.eigenstates).index(one) == 0:
                return c0.tolist()
            # <(1 - n_i)(1 - n_j)> = 1 - <n_i> - <n_j> + <n_i n_j>
            return (1.0 - o0[:, None] - o0[None, :] + c0).tolist()
        p = _probabilities(state)
        m = _one_mask(state, self.one_state).astype(float)
        return ((m * p[:, None]).T @ m).tolist()


class Energy(Observable):
    _base_tag = "energy"

    def apply(self, *, state: RydState, hamiltonian: Any, **kw: Any) -> float:
        dev = hamiltonian.observe(state) if hasattr(hamiltonian, "observe") else None
        if dev is not None:
            return dev["energy"]
        return float(np.real(hamiltonian.expect(state)))


class EnergySecondMoment(Observable):
    """<H^2> through H|psi> (default_observables.py:532-580)."""

    _base_tag = "energy_second_moment"

    def apply(self, *, state: RydState, hamiltonian: HamiltonianOperator, **kw: Any) -> float:
        dev = hamiltonian.observe(state) if hasattr(hamiltonian, "observe") else None
        if dev is not None:
            return dev["energy2"]
        applied = np.asarray(hamiltonian.apply_to(state).to_qobj())  # H|psi> or H rho H
        if state.to_qobj().isket:
            return float(np.real(np.vdot(applied, applied)))
        return float(np.real(np.trace(applied)))


class EnergyVariance(Observable):
    _base_tag = "energy_variance"
    default_aggregation = "skip_warn"  # a variance is not averaged over trajectories (:503-504)

    def apply(self, *, state: RydState, hamiltonian: HamiltonianOperator, **kw: Any) -> float:
        dev = hamiltonian.observe(state) if hasattr(hamiltonian, "observe") else None
        if dev is not None:
            return dev["energy2"] - dev["energy"] ** 2
        second = EnergySecondMoment.apply(self, state=state, hamiltonian=hamiltonian)  # type: ignore[arg-type]
        return second - float(np.real(hamiltonian.expect(state))) ** 2


# ------------------------------------------------------------------- results
class AggregationMethod(enum.IntEnum):
    """pulser/backend/observable.py:79-86 (same values, so pulser's own enum compares equal)."""

    SKIP = 0
    SKIP_WARN = 1
    MEAN = 2
    BAG_UNION = 3
    MEANSTD = 4


# AggregationMethod <-> the kinds used here
_AGG_CODE = {"skip": 0, "skip_warn": 1, "density_matrix": 1, "mean": 2, "bag_union": 3, "meanstd": 4}
_AGG_KIND = {0: "skip", 1: "skip_warn", 2: "mean", 3: "bag_union", 4: "meanstd"}


class _AbstractReprEncoder(json.JSONEncoder):
    """pulser/json/abstract_repr/serializer.py:39-60."""

    def default(self, o: Any) -> Any:
        if hasattr(o, "_to_abstract_repr"):
            return o._to_abstract_repr()
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, np.integer):
            return int(o)
        if isinstance(o, np.floating):
            return float(o)
        if isinstance(o, set):
            return list(o)
        if isinstance(o, (complex, np.complexfloating)):
            o = complex(o)
            return o.real if o.imag == 0 else dict(real=o.real, imag=o.imag)
        if type(o).__module__.startswith("torch") and hasattr(o, "tolist"):
            return o.tolist()
        return json.JSONEncoder.default(self, o)


def _deserialize_complex(obj: Any) -> Any:
    """pulser/json/abstract_repr/deserializer.py:427-437."""
    if isinstance(obj, list):
        return [_deserialize_complex(e) for e in obj]
    if isinstance(obj, dict):
        if obj.keys() == {"real", "imag"}:
            return obj["real"] + 1j * obj["imag"]
        return {k: _deserialize_complex(v) for k, v in obj.items()}
    return obj


def _check_values(values: Any) -> Any:
    """pulser/backend/aggregators.py:38-77: a non-empty list; nested elements are
    numbers, lists of numbers or lists of lists of numbers."""
    if not isinstance(values, list):
        raise ValueError("Need to supply a list of values to process.")
    if values == []:
        raise ValueError("Cannot process 0 samples.")
    return values[0]


def _check_nested(elt: Any) -> None:
    if elt == [] or len(elt) == 0:
        raise ValueError("Cannot process list of empty lists.")
    if not isinstance(elt[0], (float, complex, list)):
        raise ValueError(f"Cannot process list of lists of {type(elt[0])}.")
    if isinstance(elt[0], list):
        if len(elt[0]) == 0:
            raise ValueError("Cannot process list of matrices with empty columns.")
        if not isinstance(elt[0][0], (float, complex)):
            raise ValueError(f"Cannot process list of matrices of {type(elt[0][0])}.")


def _is_torch_tensor(x: Any) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "dim")


def _mean_of(values: list[Any]) -> Any:
    """pulser/backend/aggregators.py:119-156."""
    elt = _check_values(values)
    if _is_torch_tensor(elt):
        import torch

        return torch.stack(values).mean(dim=0)
    if isinstance(elt, np.ndarray):
        return np.stack(values).mean(axis=0)
    if isinstance(elt, (float, int)) and not isinstance(elt, bool):
        return float(np.mean(values))
    if isinstance(elt, complex):
        return complex(np.mean(values))
    if not isinstance(elt, (list, tuple)):
        raise ValueError(f"Mean aggregator cannot process data of type {type(elt)}.")
    _check_nested(elt)
    return list(np.mean(values, axis=0).tolist())


def _std_of(values: list[Any]) -> Any:
    """pulser/backend/aggregators.py:80-116 (sample standard deviation, ddof=1)."""
    elt = _check_values(values)
    if _is_torch_tensor(elt):
        import torch

        return torch.stack(values).std(dim=0)
    if isinstance(elt, np.ndarray):
        return np.stack(values).std(axis=0, ddof=1)
    if isinstance(elt, (float, int)) and not isinstance(elt, bool):
        return float(np.std(values, ddof=1))
    if isinstance(elt, complex):
        return complex(np.std(values, ddof=1))
    if not isinstance(elt, (list, tuple)):
        raise ValueError(f"Std aggregator cannot process data of type {type(elt)}.")
    _check_nested(elt)
    return list(np.std(values, axis=0, ddof=1).tolist())


class Results:
    """pulser/backend/results.py:52-490 (storage, lookup, aggregation, JSON
    abstract representation)."""

    # -- abstract representation (results.py:267-330) --------------------------
    @staticmethod
    def _encoder() -> type:
        return _AbstractReprEncoder

    def _to_abstract_repr(self) -> dict[str, Any]:
        return {
            "atom_order": [str(q) for q in self.atom_order],
            "total_duration": self.total_duration,
            "tagmap": {k: str(v) for k, v in self._tagmap.items()},
            "results": {str(k): v for k, v in self._results.items()},
            "times": {str(k): v for k, v in self._times.items()},
            "aggregation_methods": {str(k): _AGG_CODE[v] for k, v in self._aggregation.items()},
        }

    def to_abstract_repr(self, skip_validation: bool = False) -> str:
        """JSON string; numpy arrays become lists, complex numbers
        ``{"real", "imag"}`` (real numbers when the imaginary part is 0).  The
        reference validates against its JSON schema; here the document is checked
        by deserialising it again unless ``skip_validation``."""
        text = json.dumps(self._to_abstract_repr(), cls=_AbstractReprEncoder)
        if not skip_validation:
            type(self).from_abstract_repr(text)
        return text

    @classmethod
    def _from_abstract_repr(cls, obj: Mapping[str, Any]) -> "Results":
        for key in ("atom_order", "total_duration", "tagmap", "results", "times"):
            if key not in obj:
                raise ValueError(f"Invalid results representation: missing {key!r}.")
        out = cls(tuple(obj["atom_order"]), obj["total_duration"])
        for tag, uid in obj["tagmap"].items():
            out._tagmap[tag] = uuid.UUID(uid)
        for uid, value in obj["results"].items():
            out._results[uuid.UUID(uid)] = _deserialize_complex(value)
        for uid, value in obj["times"].items():
            out._times[uuid.UUID(uid)] = value
        for uid, code in obj.get("aggregation_methods", {}).items():
            out._aggregation[uuid.UUID(uid)] = _AGG_KIND[int(code)]
        return out

    @classmethod
    def from_abstract_repr(cls, repr: str) -> "Results":
        return cls._from_abstract_repr(json.loads(repr))

    def __init__(self, atom_order: tuple, total_duration: int) -> None:
        self.atom_order = tuple(atom_order)
        self.total_duration = int(total_duration)
        self._results: dict[uuid.UUID, list[Any]] = {}
        self._times: dict[uuid.UUID, list[float]] = {}
        self._tagmap: dict[str, uuid.UUID] = {}
        self._aggregation: dict[uuid.UUID, str] = {}

    def _rekey(self, tag_to_uuid: Mapping[str, uuid.UUID]) -> "Results":
        """The same results under the uuids of ANOTHER set of observables with the same tags
        (sharded runs: every rank built its own observables; tags are unique inside a configuration)."""
        for tag, new_uid in tag_to_uuid.items():
            old = self._tagmap.get(tag)
            if old is None or old == new_uid:
                continue
            for table in (self._results, self._times, self._aggregation):
                if old in table:
                    table[new_uid] = table.pop(old)
            self._tagmap[tag] = new_uid
        return self

    def _store(self, *, observable: Observable, time: float, value: Any) -> None:
        uid = observable._uuid
        self._tagmap[observable.tag] = uid
        self._aggregation[uid] = observable.default_aggregation
        if uid not in self._results:
            self._results[uid], self._times[uid] = [], []
        if time in self._times[uid]:
            raise RuntimeError(f"A value is already stored for observable '{observable.tag}' at time {time}.")
        self._results[uid].append(value)
        self._times[uid].append(time)

    def _find_uuid(self, observable: Observable | str) -> uuid.UUID:
        if isinstance(observable, Observable):
            if observable._uuid not in self._results:
                raise ValueError(f"'{observable!r}' has not been stored in the results")
            return observable._uuid
        if observable not in self._tagmap:
            raise ValueError(f"{observable!r} is not an Observable instance nor a known observable tag in the results.")
        return self._tagmap[observable]

    def get_result_tags(self) -> list[str]:
        return list(self._tagmap.keys())

    def get_result_times(self, observable: Observable | str) -> list[float]:
        return self._times[self._find_uuid(observable)]

    def get_tagged_results(self) -> dict[str, list[Any]]:
        return {tag: list(self._results[uid]) for tag, uid in self._tagmap.items()}

    def get_result(self, observable: Observable | str, time: float) -> Any:
        uid = self._find_uuid(observable)
        tol = 0.5 / self.total_duration if self.total_duration else 1e-6
        times = np.asarray(self._times[uid])
        hit = np.where(np.abs(times - time) <= tol)[0]
        if len(hit) == 0:
            raise ValueError(f"{observable!r} is not available at time {time}.")
        return self._results[uid][int(hit[0])]

    _SAMPLED_RESULT_ATTRS = ("sampling_dist", "sampling_errors", "get_samples", "get_state",
                             "plot_histogram", "n_samples", "evaluation_time", "meas_basis")

    def __getattr__(self, name: str) -> Any:
        """results.py:156-174: ``results.<tag>`` = the list of stored values."""
        if not name.startswith("_") and name in self.__dict__.get("_tagmap", {}):
            return list(self._results[self._tagmap[name]])
        if name == "bitstring_counts":
            warnings.warn("'bitstring_counts' is an attribute of the deprecated `SampledResult` class. "
                          "Please favor acessing the bitstrings via 'final_bitstrings' instead.",
                          category=FutureWarning, stacklevel=2)
            return self.final_bitstrings
        if name in self._SAMPLED_RESULT_ATTRS:
            raise AttributeError(f"{name} is available only in 'SampledResult', which has been"
                                 " deprecated and is being phased out.")
        raise AttributeError(f"{name!r} is not in the results.")

    @property
    def final_bitstrings(self) -> Counter:
        """results.py:176-190: the 'bitstrings' observable at t = 1."""
        try:
            return self.get_result("bitstrings", 1.0)
        except ValueError:
            raise RuntimeError(
                "The final bitstrings are not available. Please make sure 'BitStrings()' at relative "
                "time t=1.0 is included in the observables of your emulator backend's configuration "
                "(when possible).") from None

    @property
    def final_state(self) -> RydState:
        """results.py:192-204: the 'state' observable at t = 1."""
        try:
            return self.get_result("state", 1.0)
        except ValueError:
            raise RuntimeError(
                "The final state is not available. Please make sure 'StateResult()' at relative "
                "time t=1.0 is included in the observables of your emulator backend's configuration "
                "(when possible).") from None

    @classmethod
    def from_final_bitstrings(cls, atom_order: Sequence[str], total_duration: int,
                              final_bitstrings: Mapping[str, int]) -> "Results":
        """results.py:77-112: a Results holding only final bitstrings (e.g. from a QPU)."""
        try:
            bitstrings = Counter(final_bitstrings)
        except TypeError:
            raise TypeError("'final_bitstrings' is not a valid bitstrings counter; "
                            f"got {final_bitstrings}") from None
        obs = BitStrings(num_shots=sum(bitstrings.values()))
        obs._uuid = uuid.UUID("00000000-0000-0000-0000-000000000000")
        res = cls(tuple(atom_order), total_duration)
        res._store(observable=obs, time=1.0, value=bitstrings)
        return res

    def __str__(self) -> str:
        times = {tag: self._times[uid] for tag, uid in self._tagmap.items()}
        name = self.__class__.__name__
        return "\n".join([name, "-" * len(name), f"Stored results: {self.get_result_tags()}",
                          f"Evaluation times per result: {times}",
                          f"Atom order in states and bitstrings: {self.atom_order}",
                          f"Total sequence duration: {self.total_duration} ns"])

    @classmethod
    def aggregate(cls, results: Sequence["Results"],
                  **aggregators: Any) -> "Results":
        """results.py:331-488: combine the results of several runs tag by tag
        with each observable's default aggregation (mean, Counter union, mean and
        standard deviation, mean of density matrices) or the function / kind
        name given for the tag; 'skip' / 'skip_warn' tags are dropped."""
        if len(results) == 0:
            raise ValueError("No results to aggregate.")
        first = results[0]
        if len(results) == 1:
            return first
        common = [t for t in first._tagmap if all(t in r._tagmap for r in results)]
        for r in results:
            if r._results and not r._aggregation:
                raise NotImplementedError("You're trying to aggregate results from pulser<1.6,"
                                          "aggregation is not supported in this case.")
            for tag, uid in r._tagmap.items():
                if tag not in common and r._aggregation[uid] not in ("skip", "skip_warn"):
                    raise ValueError("You're trying to aggregate incompatible results: "
                                     f"result `{tag}` is not present in all results, "
                                     "but it's not marked to be skipped.")
        if not all({t: r._aggregation[r._tagmap[t]] for t in common}
                   == {t: first._aggregation[first._tagmap[t]] for t in common}
                   for r in results):
            raise ValueError("You're trying to aggregate incompatible results: "
                             "they do not all contain the same aggregation functions.")
        if not all(r.atom_order == first.atom_order for r in results):
            raise ValueError("You're trying to aggregate incompatible results: "
                             "they do not all have the same atom order.")
        if not all(r.total_duration == first.total_duration for r in results):
            raise ValueError("You're trying to aggregate incompatible results: "
                             "they do not all have the same sequence duration.")
        out = cls(first.atom_order, first.total_duration)
        for tag in common:
            uid = first._tagmap[tag]
            kind = first._aggregation[uid]
            agg = aggregators.get(tag, kind)
            if isinstance(agg, int) and not callable(agg):
                agg = _AGG_KIND[int(agg)]
            if agg in ("skip", "skip_warn"):
                if agg == "skip_warn":
                    warnings.warn(f"Skipping aggregation of `{tag}`.")
                continue
            times = first._times[uid]
            if not all(r._times[r._tagmap[tag]] == times for r in results):
                raise ValueError("The Results come from incompatible simulations: "
                                 f"the times for `{tag}` are not all the same.")
            uids = {r._tagmap[tag] for r in results}
            new_uid = uid if len(uids) == 1 else uuid.uuid4()
            vals_t = []
            for i in range(len(times)):
                vals = [r._results[r._tagmap[tag]][i] for r in results]
                if callable(agg):
                    vals_t.append(agg(vals))
                elif agg == "bag_union":
                    vals_t.append(sum(map(Counter, vals), Counter()))
                elif agg == "density_matrix":
                    vals_t.append(density_matrix_aggregator(vals))
                elif agg == "meanstd":
                    vals_t.append((_mean_of(vals), _std_of(vals)))
                elif agg == "mean":
                    vals_t.append(_mean_of(vals))
                else:
                    raise ValueError(f"Unknown aggregation {agg!r} for `{tag}`.")
            out._tagmap[tag] = new_uid
            out._aggregation[new_uid] = kind
            out._times[new_uid] = list(times)
            out._results[new_uid] = vals_t
        return out


# aggregated density matrices from this size on stay on the GPU (11 two-level atoms: 64 MiB)
_DEVICE_RESULT_BYTES = 64 << 20


def density_matrix_aggregator(values: Sequence[RydState]) -> RydState:
    """pulser_simulation/aggregators.py:20-37: mean of |psi><psi| (or of rho) over the trajectories.

    Formed on the device: the kets go up as one [n, D] batch and ``ryd_outer_accumulate_dim`` (fp64
    matrix cores, upper-triangle tiles) adds ``(1/n) sum |psi><psi|``; trajectory density matrices
    are added elementwise (``ryd_accumulate``).  No D x D array is built on the host; results of
    64 MiB and more are handed back as a :class:`DeviceState` (copied on explicit request only)."""
    from .engine import _torch, accumulate, outer_accumulate

    if len(values) == 0:
        raise ValueError("Cannot average an empty list of states.")
    torch = _torch()  # (raises "pulser_amd needs an AMD GPU ... there is no CPU fallback" on a host without one)
    dev = torch.device("cuda", torch.cuda.current_device())
    D = int(values[0].to_qobj().shape[0])
    w = 1.0 / len(values)
    acc = torch.zeros((D, D), dtype=torch.complex128, device=dev)
    kets = [st.to_qobj() for st in values if st.to_qobj().isket]
    chunk = max(1, (1 << 30) // (16 * D))  # <= 1 GiB of kets per upload
    for lo in range(0, len(kets), chunk):
        part = kets[lo:lo + chunk]
        host = np.empty((len(part), D), dtype=np.complex128)
        for i, k in enumerate(part):
            host[i] = np.asarray(k).reshape(-1)
        outer_accumulate(torch.from_numpy(host).to(dev), acc, np.full(len(part), w))
    for st in values:
        s = st.to_qobj()
        if s.isket:
            continue
        t = getattr(s, "device_tensor", None)
        if t is None:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(s), dtype=np.complex128)).to(dev)
        accumulate(t.contiguous(), acc, w)
    if 16 * D * D >= _DEVICE_RESULT_BYTES:
        return RydState(DeviceState(tensor=acc), eigenstates=values[0].eigenstates)
    return RydState(acc.cpu().numpy(), eigenstates=values[0].eigenstates)


# -------------------------------------------------------------------- config
class QutipConfig:
    """``QutipConfig`` / ``EmulationConfig`` (qutip_config.py:28-192,
    pulser/backend/config.py:151-470): the options this backend consumes."""

    def __init__(self, *, observables: Sequence[Observable] = (), callbacks: Sequence[Callable] = (),
                 default_evaluation_times: Any = (1.0,), initial_state: RydState | None = None,
                 with_modulation: bool = False, noise_model: Any = None,
                 prefer_device_noise_model: bool = False, n_trajectories: int | None = None,
                 sampling_rate: float = 1.0, solver: Solver = Solver.DEFAULT,
                 default_num_shots: int = 1000, progress_bar: bool = False,
                 print_progress: bool = False, **backend_options: Any) -> None:
        if unexpected := set(backend_options) - {"interaction_matrix"}:  # config.py:80-90
            raise ValueError(
                f"'QutipConfig' received unexpected keyword arguments: {unexpected}; only the "
                f"following keyword arguments are expected: {self._expected_kwargs()}. "
            )
        if initial_state and not isinstance(initial_state, RydState):
            raise TypeError(
                "If provided, `initial_state` must be an instance of "
                f"`RydState`, not {type(initial_state)}."
            )
        if noise_model is not None and getattr(noise_model, "samples_per_run", None) not in (None, 1):
            warnings.warn(
                f"The number of samples per run (`samples_per_run` = {noise_model.samples_per_run}) "
                "is ignored when using QutipBackendV2.", stacklevel=2)
        try:
            solver = Solver(solver)
        except ValueError:
            raise ValueError(f"Invalid solver '{solver}'. Allowed solvers are: "
                             + ", ".join(v.value for v in Solver) + ".") from None
        if not observables and not callbacks:  # config.py:259-266
            warnings.warn("'QutipConfig' was initialized without any observables. The corresponding "
                          "emulation results will be empty.", stacklevel=2)
        for i, cb in enumerate(callbacks):
            if isinstance(cb, Observable):
                raise TypeError("All entries in 'callbacks' must not be instances of Observable, since "
                                f"those go in 'observables'. Instead, got {cb!r} at index {i}.")
            if not callable(cb):
                raise TypeError("All entries in 'callbacks' must be instances of Callback. Instead, got "
                                f"instance of type {type(cb)} at index {i}: {cb!r}.")
        for i, obs in enumerate(observables):
            if not isinstance(obs, Observable):
                raise TypeError("All entries in 'observables' must be instances of Observable. Instead, "
                                f"got instance of type {type(obs)} at index {i}: {obs!r}.")
        tags = [o.tag for o in observables]
        if repeated := [k for k, v in Counter(tags).items() if v > 1]:
            raise ValueError(
                "Some of the provided 'observables' share identical tags. Use 'tag_suffix' when "
                "instantiating multiple instances of the same observable so they can be distinguished. "
                f"Repeated tags found: {repeated}")
        self.observables = tuple(observables)
        self.callbacks = tuple(callbacks)
        if isinstance(default_evaluation_times, str):
            if default_evaluation_times != "Full":
                raise ValueError(f"'default_evaluation_times' must be 'Full' or a sequence of floats, not {default_evaluation_times!r}.")
            self.default_evaluation_times: Any = "Full"
        else:
            self.default_evaluation_times = _validate_eval_times(list(map(float, default_evaluation_times)))
        self.initial_state = initial_state
        self.with_modulation = bool(with_modulation)
        if noise_model is None:
            noise_model = NoiseModel()
        elif not hasattr(noise_model, "noise_types"):  # pulser.NoiseModel instances are welcome
            raise TypeError(f"When defined, 'noise_model' must be a NoiseModel instance, not {type(noise_model)}.")
        self.noise_model = noise_model
        self.prefer_device_noise_model = bool(prefer_device_noise_model)
        runs = getattr(noise_model, "runs", None)
        if n_trajectories is not None and runs is not None and n_trajectories != runs:
            raise ValueError("`EmulationConfig.n_trajectories` and `NoiseModel.runs` can't be simultaneously "
                             "defined. Please favour using only `EmulationConfig.n_trajectories`.")
        if n_trajectories is None:  # config.py:368-374
            n_trajectories = 40 if prefer_device_noise_model else (runs if runs is not None else 1)
        if n_trajectories < 1 or n_trajectories != int(n_trajectories):
            raise ValueError(f"`n_trajectories` must be a strictly positive integer, not {n_trajectories}.")
        self.n_trajectories = int(n_trajectories)
        if not (0 < sampling_rate <= 1.0):
            raise ValueError(f"The sampling rate (`sampling_rate` = {sampling_rate}) must be greater than 0 and less than or equal to 1.")
        self.sampling_rate = sampling_rate
        self.solver = Solver(solver)
        self.default_num_shots = int(default_num_shots)
        self.progress_bar = progress_bar
        self.print_progress = print_progress
        if backend_options.get("interaction_matrix") is not None:
            raise NotImplementedError("'QutipBackendV2' does not handle custom interaction matrices.")
        self._extra = dict(backend_options)

    state_type = RydState  # config.py:409-417
    operator_type = RydOperator

    @staticmethod
    def _expected_kwargs() -> set[str]:
        """config.py:396-407 + qutip_config.py:143-149."""
        return {"callbacks", "observables", "default_evaluation_times", "initial_state",
                "with_modulation", "interaction_matrix", "prefer_device_noise_model", "noise_model",
                "n_trajectories", "sampling_rate", "solver", "print_progress", "progress_bar"}

    # -- read-only views / JSON abstract representation (config.py:116-118, 438-470)
    @property
    def _backend_options(self) -> dict[str, Any]:
        ev = self.default_evaluation_times
        return {
            "sampling_rate": self.sampling_rate, "solver": self.solver,
            "print_progress": self.print_progress, "progress_bar": self.progress_bar,
            "callbacks": list(self.callbacks), "observables": list(self.observables),
            "default_evaluation_times": ev if isinstance(ev, str) else ev.tolist(),
            "initial_state": self.initial_state, "with_modulation": self.with_modulation,
            "interaction_matrix": None, "prefer_device_noise_model": self.prefer_device_noise_model,
            "noise_model": self.noise_model, "n_trajectories": self.n_trajectories,
            "default_num_shots": self.default_num_shots,
        }

    def with_changes(self, **changes: Any) -> "QutipConfig":
        """A copy of the configuration with the given changes."""
        opts = self._backend_options | changes
        opts.pop("interaction_matrix", None)
        return type(self)(**opts)

    def _to_abstract_repr(self) -> dict[str, Any]:
        d = self._backend_options
        if d["callbacks"]:
            raise ValueError("Callbacks cannot be serialized.")
        d["solver"] = self.solver.value
        return d

    def to_abstract_repr(self, skip_validation: bool = False) -> str:
        """JSON document of the configuration; ``from_abstract_repr`` reads it
        back (and reads the documents pulser-core's ``EmulationConfig`` writes)."""
        text = json.dumps(self, cls=_AbstractReprEncoder)
        if not skip_validation:
            type(self).from_abstract_repr(text)
        return text

    @classmethod
    def from_abstract_repr(cls, obj_str: str) -> "QutipConfig":
        """json/abstract_repr/deserializer.py (``_deserialize_emulation_config``)."""
        if not isinstance(obj_str, str):
            raise TypeError("The serialized EmulationConfig must be given as a string. "
                            f"Instead, got object of type {type(obj_str)}.")
        obj = json.loads(obj_str)
        kinds = {c._base_tag: c for c in (BitStrings, Occupation, CorrelationMatrix, Energy,
                                          EnergyVariance, EnergySecondMoment, Fidelity, Expectation)}
        observables = []
        for o in obj.get("observables", []):
            if o["observable"] not in kinds:
                raise ValueError(f"Observable {o['observable']!r} cannot be deserialized.")
            kw: dict[str, Any] = {"evaluation_times": o.get("evaluation_times"),
                                  "tag_suffix": o.get("tag_suffix")}
            if "num_shots" in o:
                kw["num_shots"] = o["num_shots"]
            if "one_state" in o:
                kw["one_state"] = o["one_state"]
            if o["observable"] == "expectation":
                op = o["operator"]
                ops = [(_deserialize_complex(c), [(_deserialize_complex(q), set(inds)) for q, inds in t])
                       for c, t in op["operations"]]
                observables.append(Expectation(RydOperator.from_operator_repr(
                    eigenstates=tuple(op["eigenstates"]), n_qudits=op["n_qudits"], operations=ops), **kw))
            elif "state" in o:
                st = o["state"]
                amps = {k: (v["real"] + 1j * v["imag"] if isinstance(v, dict) else v)
                        for k, v in st["amplitudes"].items()}
                observables.append(Fidelity(RydState.from_state_amplitudes(
                    eigenstates=tuple(st["eigenstates"]), amplitudes=amps), **kw))
            else:
                observables.append(kinds[o["observable"]](**kw))
            if "uuid" in o:  # optional in the schema
                observables[-1]._uuid = uuid.UUID(o["uuid"])
            if "default_aggregation_method" in o:
                observables[-1].default_aggregation = _AGG_KIND[int(o["default_aggregation_method"])]
        init = obj.get("initial_state")
        if init is not None:
            amps = {k: (v["real"] + 1j * v["imag"] if isinstance(v, dict) else v)
                    for k, v in init["amplitudes"].items()}
            init = RydState.from_state_amplitudes(eigenstates=tuple(init["eigenstates"]), amplitudes=amps)
        nm = obj.get("noise_model")
        kwargs = dict(
            observables=observables, default_evaluation_times=obj.get("default_evaluation_times", (1.0,)),
            initial_state=init, with_modulation=obj.get("with_modulation", False),
            noise_model=NoiseModel._from_abstract_repr(nm) if nm is not None else None,
            prefer_device_noise_model=obj.get("prefer_device_noise_model", False),
            n_trajectories=obj.get("n_trajectories"), interaction_matrix=obj.get("interaction_matrix"),
        )
        for key in ("sampling_rate", "solver", "print_progress", "progress_bar"):
            if key in obj:
                kwargs[key] = obj[key]
        if obj.get("default_num_shots") is not None:
            kwargs["default_num_shots"] = obj["default_num_shots"]
        return cls(**kwargs)

    def is_evaluation_time(self, t: float, tol: float = 1e-6) -> bool:
        """config.py:419-427."""
        if isinstance(self.default_evaluation_times, str):
            return 0.0 <= t <= 1.0
        return self.is_time_in_evaluation_times(t, self.default_evaluation_times, tol=tol)

    @staticmethod
    def is_time_in_evaluation_times(t: float, evaluation_times: Any, tol: float = 1e-6) -> bool:
        """config.py:429-436."""
        return 0.0 <= t <= 1.0 and bool(
            np.any(np.abs(np.array(evaluation_times, dtype=float) - t) <= tol))

    def _get_legacy_evaluation_times(self, total_duration_ns: int) -> Any:
        """qutip_config.py:169-192."""
        if self.callbacks:
            return "Full"
        extra: set[float] = set()
        for obs in self.observables:
            if obs.evaluation_times is not None:
                extra.update(obs.evaluation_times)
        rel = self.default_evaluation_times
        if extra:
            if isinstance(rel, str):
                rel = np.linspace(0, total_duration_ns - 1,
                                  int(self.sampling_rate * total_duration_ns), dtype=int) / total_duration_ns
            rel = np.union1d(rel, list(extra))
        return "Full" if isinstance(rel, str) else rel * total_duration_ns * 1e-3


# ------------------------------------------------------------- legacy backend
class EmulatorConfig:
    """``pulser.EmulatorConfig`` (pulser/backend/config.py:477-583): the options of the
    legacy ``QutipBackend``."""

    _EVAL_LABELS = ("Full", "Minimal", "Final")

    def __init__(self, backend_options: dict[str, Any] | None = None, sampling_rate: float = 1.0,
                 evaluation_times: Any = "Full", initial_state: Any = "all-ground",
                 with_modulation: bool = False, prefer_device_noise_model: bool = False,
                 noise_model: Any = None) -> None:
        if not (0 < sampling_rate <= 1.0):
            raise ValueError(f"The sampling rate (`sampling_rate` = {sampling_rate}) must be "
                             "greater than 0 and less than or equal to 1.")
        if isinstance(evaluation_times, str):
            if evaluation_times not in self._EVAL_LABELS:
                raise ValueError("If provided as a string, 'evaluation_times' must be one "
                                 f"of the following options: {self._EVAL_LABELS}")
        elif isinstance(evaluation_times, float):
            if not (0 < evaluation_times <= 1.0):
                raise ValueError("If provided as a float, 'evaluation_times' must be"
                                 " greater than 0 and less than or equal to 1.")
        elif isinstance(evaluation_times, (list, tuple, np.ndarray)):
            if np.min(evaluation_times, initial=0) < 0:
                raise ValueError("If provided as a sequence of values, 'evaluation_times' must not "
                                 "contain negative values.")
        else:
            raise TypeError(f"'{type(evaluation_times)}' is not a valid type for 'evaluation_times'.")
        if isinstance(initial_state, str) and initial_state != "all-ground":
            raise ValueError("If provided as a string, 'initial_state' must be 'all-ground'.")
        self.backend_options = dict(backend_options or {})
        self.sampling_rate, self.evaluation_times = sampling_rate, evaluation_times
        self.initial_state, self.with_modulation = initial_state, bool(with_modulation)
        self.prefer_device_noise_model = bool(prefer_device_noise_model)
        self.noise_model = noise_model if noise_model is not None else NoiseModel()


class QutipBackend:
    """The deprecated ``QutipBackend`` (qutip_backend.py:44-118): a thin wrapper that
    builds a ``QutipEmulator`` from a legacy ``EmulatorConfig`` and returns its
    ``CoherentResults`` / ``NoisyResults``."""

    def __init__(self, sequence: Any, config: Any = None, mimic_qpu: bool = False) -> None:
        warnings.warn("'QutipBackend' is deprecated. Please use 'QutipBackendV2' instead.",
                      DeprecationWarning, stacklevel=2)
        config = EmulatorConfig() if config is None else config
        needed = ("sampling_rate", "evaluation_times", "initial_state", "with_modulation",
                  "prefer_device_noise_model", "noise_model")
        if isinstance(config, QutipConfig) or not all(hasattr(config, a) for a in needed):
            raise TypeError(f"'config' must be of type 'EmulatorConfig', not {type(config)}.")
        self._config = config
        noise_model = None
        if config.prefer_device_noise_model:
            noise_model = getattr(getattr(sequence, "device", None), "noise_model", None)
        kw = dict(sampling_rate=config.sampling_rate, noise_model=noise_model or config.noise_model,
                  evaluation_times=config.evaluation_times)
        if hasattr(sequence, "_schedule"):
            self._sim_obj = QutipEmulator.from_sequence(sequence, with_modulation=config.with_modulation, **kw)
        else:
            self._sim_obj = QutipEmulator(sequence, **kw)
        self._sim_obj.set_initial_state(config.initial_state)

    def run(self, progress_bar: bool = False, **options: Any) -> Any:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)
            return self._sim_obj.run(progress_bar=progress_bar, **options)


def register_with_pulser() -> bool:
    """When pulser-core is importable: make ``RydState`` / ``RydOperator`` virtual subclasses of
    ``pulser.backend.State`` / ``Operator`` (backend/state.py:34, backend/operator.py:38), so that
    pulser-side ``isinstance`` checks (e.g. ``EmulationConfig(initial_state=...)``,
    ``Fidelity(state)``) accept them.  Returns False when pulser is absent."""
    try:
        import pulser.backend as pb
    except Exception:
        return False
    pb.State.register(RydState)
    pb.Operator.register(RydOperator)
    return True


def _adopt_pulser_config(config: Any) -> "QutipConfig":
    """A ``pulser.backend.EmulationConfig`` (backend/config.py:151-470; what
    ``EmulatorBackend.validate_config`` hands to a backend, backend/abc.py:143-169) converted to
    this package's ``QutipConfig``: observables by tag through the JSON abstract representation
    both sides write (json/abstract_repr), callbacks carried over as objects."""
    if not (hasattr(config, "to_abstract_repr") and hasattr(config, "_backend_options")):
        raise TypeError("'config' must be an instance of 'EmulationConfig'")
    register_with_pulser()
    callbacks = list(getattr(config, "callbacks", ()) or ())
    bare = config.with_changes(callbacks=[]) if callbacks else config
    try:
        text = bare.to_abstract_repr(skip_validation=True)
    except TypeError:  # older signature
        text = bare.to_abstract_repr()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "initialized without any observables" was already issued by pulser
        mine = QutipConfig.from_abstract_repr(text)
        return mine.with_changes(callbacks=callbacks) if callbacks else mine


# ------------------------------------------------------------------- backend
class QutipBackendV2:
    """qutip_backend.py:121-325 on the MI355X engine.  ``sequence`` is a
    ``pulser.Sequence`` (needs pulser) or a ``pulser_amd.SequenceInputs``."""

    default_config: "QutipConfig"  # set right after the class (qutip_backend.py:136-138)
    config_type: type  # = QutipConfig
    last_timing: dict[str, float] | None = None
    last_observable_engine_stats: dict[str, Any] | None = None

    def __init__(self, sequence: Any, *, config: QutipConfig | None = None,
                 mimic_qpu: bool = False) -> None:
        if config is not None and not isinstance(config, QutipConfig):
            config = _adopt_pulser_config(config)
        self._config = config or self.default_config
        cfg = self._config
        nm = self._get_noise_model(cfg, getattr(sequence, "device", None))
        has_dmm = (any(c.is_dmm for c in sequence.channels) if hasattr(sequence, "channels")
                   else any(type(ch).__name__ == "DMM"
                            for ch in getattr(sequence, "declared_channels", {}).values()))
        if (nm is not None and has_dmm and "register" in nm.noise_types
                and nm.detuning_map_spot_waist is None):  # pulser/backend/abc.py:106-121
            raise ValueError(
                "Combining register noise with a DMM requires"
                "`detuning_map_spot_waist` to be defined. If not defined,"
                "atom thermal motion can lead to non-physical effects."
            )
        device_nm = getattr(getattr(sequence, "device", None), "noise_model", None)
        if (cfg.prefer_device_noise_model and device_nm is not None
                and getattr(device_nm, "runs", None) is not None
                and device_nm.runs != cfg.n_trajectories):  # pulser/backend/abc.py:123-135
            warnings.warn(f"'sequence.device.noise_model.runs={device_nm.runs}' is being ignored; "
                          f"'config.n_trajectories={cfg.n_trajectories}' will be used instead.", stacklevel=2)
        kw = dict(sampling_rate=cfg.sampling_rate, noise_model=nm, solver=cfg.solver,
                  n_trajectories=cfg.n_trajectories)
        if hasattr(sequence, "_schedule"):
            self._sim_obj = QutipEmulator.from_sequence(sequence, with_modulation=cfg.with_modulation, **kw)
        else:
            self._sim_obj = QutipEmulator(sequence, **kw)
        self._options = self._prepare(self._sim_obj, cfg)

    @staticmethod
    def _get_noise_model(cfg: QutipConfig, device: Any) -> Any:
        """qutip_backend.py: the device's noise model wins when the configuration
        prefers it and the device has one."""
        if cfg.prefer_device_noise_model and getattr(device, "noise_model", None):
            return device.noise_model
        return cfg.noise_model

    @staticmethod
    def _prepare(sim: QutipEmulator, cfg: QutipConfig) -> dict[str, Any]:
        sim.set_evaluation_times(cfg._get_legacy_evaluation_times(sim.total_duration_ns))
        if cfg.initial_state:
            sim.set_initial_state(np.asarray(cfg.initial_state.to_qobj()).reshape(-1))
        options = {"print_progress": cfg.print_progress, "progress_bar": cfg.progress_bar}
        sim._validate_options(dict(options))
        return options

    def run(self) -> Results:
        return self._run_raw(self._sim_obj, self._config, dict(self._options))

    @staticmethod
    def run_from_sequence_samples(sequence_samples: Any, register: Any = None, device: Any = None,
                                  *, config: QutipConfig | None = None) -> Results:
        """qutip_backend.py:194-232: emulate already sampled sequences (pulser
        ``SequenceSamples`` + register + device, or ``SequenceInputs``)."""
        cfg = config or QutipBackendV2.default_config
        sim = QutipEmulator(sequence_samples, register, device, sampling_rate=cfg.sampling_rate,
                            noise_model=QutipBackendV2._get_noise_model(cfg, device),
                            solver=cfg.solver, n_trajectories=cfg.n_trajectories)
        return QutipBackendV2._run_raw(sim, cfg, QutipBackendV2._prepare(sim, cfg))

    @staticmethod
    def _run_raw(sim: QutipEmulator, config: QutipConfig, options: dict[str, Any]) -> Results:
        from .engine import Engine, GeneralEngine
        from .general import lower_general

        eigenstates = sim._hamiltonian_data.eigenbasis
        qids = tuple(sim.samples_obj.qubit_ids)
        T = sim.total_duration_ns
        with_leakage = bool(getattr(config.noise_model, "with_leakage", False))
        holder: list[Any] = []  # the engine of the noiseless H(t), built on first use

        def noiseless_engine() -> Any:
            # qutip_backend.py:259-264: _get_noiseless_hamiltonian(with_leakage) is first
            # called (and, for the leakage basis, first draws its HamiltonianData) inside
            # the loop over evaluation times, i.e. after the solver ran; lru_cached after.
            # Matrix-free for 2-level Ising sequences, explicit sparse terms otherwise
            # (leakage: the states are 3^N / 4^N dimensional, so is this operator).
            if not holder:
                hd = sim._get_noiseless_data(with_leakage)
                noiseless = dict(hd.problem(hd.noise_trajectories[0], sim._sampling_rate))
                noiseless["collapse_ops"] = []
                holder.append(Engine.from_problems([noiseless], mode="sesolve")
                              if sim._fast_path_ok(noiseless)
                              else GeneralEngine(lower_general(noiseless, mesolve=False)))
            return holder[0]

        def fill(res: Results, coherent: Any) -> None:
            for r in coherent:
                t = r.evaluation_time
                state = RydState(r.state.unit(), eigenstates=eigenstates)
                ham = HamiltonianOperator(noiseless_engine(), t * T / 1000, eigenstates)
                for cb in config.callbacks:
                    cb(config=config, t=float(t), state=state, hamiltonian=ham, result=res)
                for obs in config.observables:
                    obs(config=config, t=float(t), state=state, hamiltonian=ham, result=res)

        import time as _time

        timing = {"solve_s": 0.0, "observables_s": 0.0}
        try:
            if not has_stochastic_noise(sim.noise_model):
                tic = _time.perf_counter()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore", DeprecationWarning)
                    single = sim.run(**options)
                timing["solve_s"] = _time.perf_counter() - tic
                res = Results(qids, T)
                tic = _time.perf_counter()
                fill(res, single)
                timing["observables_s"] = _time.perf_counter() - tic
                return res
            dist = sim._distributed()
            builtin = (StateResult, BitStrings, Occupation, CorrelationMatrix, Energy, EnergyVariance,
                       EnergySecondMoment, Fidelity, Expectation)
            if dist is not None and not config.callbacks and all(type(o) in builtin for o in config.observables):
                # one process per GPU: every rank replays rank 0's random stream and trajectory
                # draws, solves and observes only its block of trajectories, skips the draws of the
                # others (only BitStrings consumes random numbers), and the per-trajectory Results
                # are gathered before the aggregation - identical output for any world size
                from .distributed import partition

                rank, world = dist.get_rank(), dist.get_world_size()
                payload: list[Any] = [None]
                if rank == 0:
                    sim._refresh_trajectories_if_used()
                    payload = [(sim._hamiltonian_data.noise_trajectories, np.random.get_state())]
                dist.broadcast_object_list(payload, src=0)
                trajs, rng_state = payload[0]
                sim._hamiltonian_data.noise_trajectories = trajs
                sim._problems_cache = None
                sim._noise_trajectories_used = False  # already fresh: _noisy_runs must not redraw
                np.random.set_state(rng_state)
                lo, hi = partition([t.reps for t in trajs], world)[rank]
                rel_times = [float(t / (T * 1e-3)) for t in sim._eval_times_array]
                mine: list[tuple[int, Results]] = []
                order = 0
                tic = _time.perf_counter()
                for coherent, reps in sim._noisy_runs(only=(lo, hi), **options):
                    timing["solve_s"] += _time.perf_counter() - tic
                    tic = _time.perf_counter()
                    for _ in range(reps):
                        if coherent is None:  # another rank's trajectory: keep the stream in step
                            for t in rel_times:
                                for obs in config.observables:
                                    if isinstance(obs, BitStrings) and obs._fires(config, t, T):
                                        obs._skip_draws(config, len(qids))
                        else:
                            res = Results(qids, T)
                            fill(res, coherent)
                            mine.append((order, res))
                        order += 1
                    timing["observables_s"] += _time.perf_counter() - tic
                    tic = _time.perf_counter()
                gathered: list[Any] = [None] * world
                dist.all_gather_object(gathered, mine)
                mine_tags = {o.tag: o._uuid for o in config.observables}
                ordered = [r._rekey(mine_tags) for _, r in
                           sorted((x for part in gathered for x in part), key=lambda kv: kv[0])]
                return Results.aggregate(ordered)
            results: list[Results] = []
            tic = _time.perf_counter()
            for coherent, reps in sim._noisy_runs(**options):
                timing["solve_s"] += _time.perf_counter() - tic
                tic = _time.perf_counter()
                for _ in range(reps):
                    res = Results(qids, T)
                    fill(res, coherent)
                    results.append(res)
                timing["observables_s"] += _time.perf_counter() - tic
                tic = _time.perf_counter()
            return Results.aggregate(results)
        finally:
            # diagnostics of the last run: wall-clock split and the launches of the noiseless-H engine
            QutipBackendV2.last_timing = timing
            QutipBackendV2.last_observable_engine_stats = holder[0].stats() if holder else None
            for eng in holder:
                eng.close()


# the name the INTEGRATION.md registry entry resolves (pulser/backends.py:49-58: getattr(module, name))
RydEmuBackend = QutipBackendV2

QutipBackendV2.default_config = QutipConfig(observables=[BitStrings(evaluation_times=[1.0]), StateResult()])
QutipBackendV2.config_type = QutipConfig
