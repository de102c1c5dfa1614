"""Result marshalling and bitstring sampling (host side, NumPy replay).

Mirrors ``pulser_simulation.qutip_result.QutipResult``,
``pulser.result.Result/SampledResult`` and
``pulser_simulation.simresults.{SimulationResults,CoherentResults,NoisyResults}``.
States come back from the GPU as NumPy arrays; the sampling replays the
reference's exact NumPy call sequence on the global ``np.random`` stream, which
is what makes sampled bitstring indices bit-exact:

* weights: ``np.abs(state)**2`` (or ``|diag rho|``), reversed for ground-rydberg,
  normalised by a sequential sum (pulser_simulation/qutip_result.py:101-158);
* ``multinomial``: ``rand(n)`` then ``searchsorted(cumsum(p))``
  (pulser-core/pulser/math/multinomial.py:18-36);
* SPAM measurement flips (pulser_simulation/simresults.py:537-568);
* first-match time lookup (simresults.py:176-190).
"""

from __future__ import annotations

import collections.abc
import os
import threading
import warnings
import weakref
from collections import Counter
from dataclasses import dataclass
from functools import lru_cache
from typing import Any, Mapping, Optional, Sequence

import numpy as np

EIGENSTATES = {"ground-rydberg": ["r", "g"], "digital": ["g", "h"], "XY": ["u", "d"]}
_ONE_STATE = {"ground-rydberg": "r", "digital": "h", "XY": "d"}


class QState(np.ndarray):
    """A ket (shape (D, 1)) or density matrix (D, D) with the few ``qutip.Qobj``
    accessors user code of the reference relies on (``full``, ``isket``,
    ``norm``, ``unit``, ``overlap``, ``dag``, ``diag``, ``tr``)."""

    def __new__(cls, data: Any) -> "QState":
        arr = np.asarray(data, dtype=complex)
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        return arr.view(cls)

    @property
    def isket(self) -> bool:
        return self.shape[1] == 1 and self.shape[0] > 1 or self.shape == (1, 1)

    @property
    def isoper(self) -> bool:
        return not self.isket

    def full(self) -> np.ndarray:
        return np.array(self)

    def dag(self) -> "QState":
        return QState(np.conj(np.asarray(self)).T)

    def diag(self) -> np.ndarray:
        return np.diag(np.asarray(self))

    def tr(self) -> complex:
        return complex(np.trace(np.asarray(self)))

    def norm(self) -> float:
        a = np.asarray(self)
        if self.isket:
            return float(np.linalg.norm(a))
        return float(np.sum(np.linalg.svd(a, compute_uv=False)))

    def unit(self) -> "QState":
        return QState(np.asarray(self) / self.norm())

    def overlap(self, other: Any) -> complex:
        """``Qobj.overlap``: <a|b>, <a|B|a>, <b|A|b> or Tr(A^dag B)."""
        other = QState(other)
        a, b = np.asarray(self), np.asarray(other)
        if self.isket and other.isket:
            return complex(np.vdot(a, b))
        if self.isket:
            return complex(np.vdot(a, b @ a))
        if other.isket:
            return complex(np.vdot(b, a @ b))
        return complex(np.trace(a.conj().T @ b))


class DeviceState:
    """A density matrix that stays on the GPU (mesolve at 13+ atoms: 1 - 4 GiB per state).

    What the results of the reference need from ``qutip.Qobj`` on the sampling path - ``shape``,
    ``isket``, ``diag()`` (qutip_result.py:101-118) - is served from the diagonal, which is
    reduced on the device and is all that ever crosses PCIe by itself.  ``full()`` /
    ``np.asarray`` copy the whole matrix to the host on explicit request.  The initial state
    of a run (|psi><psi| of a ket) is kept as its ket and expanded on demand only.
    """

    isket = False
    isoper = True

    def __init__(self, tensor: Any = None, ket: np.ndarray | None = None) -> None:
        if (tensor is None) == (ket is None):
            raise ValueError("give either a device tensor or a ket")
        self._tensor = tensor
        self._ket = None if ket is None else np.asarray(ket, dtype=complex).reshape(-1)
        self._diag: np.ndarray | None = None

    @property
    def shape(self) -> tuple[int, int]:
        d = int(self._tensor.shape[-1]) if self._tensor is not None else int(self._ket.size)
        return (d, d)

    @property
    def device_tensor(self) -> Any:
        """The torch tensor on the GPU (None for the expanded-ket form)."""
        return self._tensor

    def diag(self) -> np.ndarray:
        if self._diag is None:
            if self._tensor is not None:
                import torch

                self._diag = torch.diagonal(self._tensor, dim1=-2, dim2=-1).contiguous().cpu().numpy()
            else:
                self._diag = (np.abs(self._ket) ** 2).astype(complex)
        return self._diag

    def tr(self) -> complex:
        return complex(np.sum(self.diag()))

    def full(self) -> np.ndarray:
        if self._tensor is not None:
            return self._tensor.cpu().numpy()
        return np.outer(self._ket, self._ket.conj())

    def copy(self) -> np.ndarray:
        return self.full()

    def __array__(self, dtype: Any = None, copy: Any = None) -> np.ndarray:
        a = self.full()
        return a if dtype is None else a.astype(dtype)

    def dag(self) -> "QState":
        return QState(self.full()).dag()

    def norm(self) -> float:
        return QState(self.full()).norm()

    def unit(self) -> "QState":
        return QState(self.full()).unit()

    def overlap(self, other: Any) -> complex:
        return QState(self.full()).overlap(other)

    def __repr__(self) -> str:
        where = "device tensor" if self._tensor is not None else "ket form"
        return f"DeviceState(shape={self.shape}, {where})"


class SnapshotStore:
    """The evaluation-time snapshots of one solve, ``[n_times - 1, B, dim...]`` on the GPU, and their host copies.

    The reference hands every state of ``result.states`` to the host (simulation.py:744-748); with
    ``evaluation_times="Full"`` - its default - that is 3 101 x 256 KiB = 813 MB for one 14-atom sequence, most of which
    nobody reads.  The snapshots stay in HBM; a state crosses PCIe when it is read (``LazyState``), and once more than
    ``bulk_after`` different states have been asked for the whole tensor is copied in one transfer (a loop over all
    evaluation times - ``expect``, sampling at every time - then costs one copy, as before).

    **Retention is bounded.**  Results objects a sweep keeps around would otherwise pin 813 MB of HBM each: every live
    store is registered (weakly), and before a new one is added the OLDEST stores are spilled to the host until the
    retained device bytes fit ``device_budget_bytes()`` (default 1/8 of the device's memory; ``PULSER_AMD_SNAPSHOT_GB``
    overrides).  Spilling changes where a state is read from, never what is read.  ``spill_all()`` / a results object's
    ``to_host()`` move everything explicitly."""

    _live: "list[weakref.ref[SnapshotStore]]" = []   # registration order = age
    _lock = threading.Lock()

    def __init__(self, tensor: Any, bulk_after: int = 16) -> None:
        self._dev = tensor
        self._host: list[np.ndarray] | None = None  # after fetch_all: one array per evaluation time
        self._reads = 0
        self._bulk_after = int(bulk_after)
        self._register()

    # -- the registry -------------------------------------------------------------------------------------------------
    @staticmethod
    def device_budget_bytes() -> int:
        env = os.environ.get("PULSER_AMD_SNAPSHOT_GB")
        if env is not None:
            return int(float(env) * 2**30)
        try:
            import torch

            return int(torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory // 8)
        except Exception:  # no device (CPU tests build stores over host tensors)
            return 8 * 2**30

    def device_bytes(self) -> int:
        dev = self._dev
        return 0 if dev is None else int(getattr(dev, "nbytes", 0))

    @classmethod
    def _alive(cls) -> "list[SnapshotStore]":
        stores = [r() for r in cls._live]
        cls._live = [r for r, s in zip(cls._live, stores) if s is not None and s._dev is not None]
        return [s for s in stores if s is not None and s._dev is not None]

    @classmethod
    def retained_device_bytes(cls) -> int:
        with cls._lock:
            return sum(s.device_bytes() for s in cls._alive())

    def _register(self) -> None:
        cls = type(self)
        with cls._lock:
            budget = cls.device_budget_bytes()
            alive = cls._alive()
            held = sum(s.device_bytes() for s in alive) + self.device_bytes()
            for s in alive:  # oldest first
                if held <= budget:
                    break
                held -= s.device_bytes()
                s.fetch_all()
            cls._live = [r for r in cls._live if (r() is not None and r()._dev is not None)]
            cls._live.append(weakref.ref(self))

    @classmethod
    def spill_all(cls) -> int:
        """Move every live store to the host (frees their HBM); returns the bytes moved."""
        with cls._lock:
            moved = 0
            for s in cls._alive():
                moved += s.device_bytes()
                s.fetch_all()
            cls._live = []
            return moved

    # -- reads ----------------------------------------------------------------------------------------------------------
    @property
    def device_tensor(self) -> Any:
        """The torch tensor on the GPU (None once everything has been copied to the host)."""
        return self._dev

    def fetch_all(self) -> "list[np.ndarray]":
        """Everything to the host: ONE array per evaluation time ([B, dim...]), not one array for the run - a state that
        outlives the run then keeps its own time slice alive and nothing else, without a second copy (round 6: copying
        every state out of one big host array cost 220 ms of page faults when all 3 101 states of a 14-atom run were read)."""
        if self._host is None:
            dev = self._dev
            self._host = [dev[i].cpu().numpy() for i in range(int(dev.shape[0]))]
            self._dev = None
        return self._host

    def get(self, i: int, b: int) -> np.ndarray:
        if self._host is None:
            self._reads += 1
            if self._reads <= self._bulk_after:
                return self._dev[i, b].cpu().numpy()
            self.fetch_all()
        slab = self._host[i]
        # one sequence per run: the time slice IS the state (a view of its own array); batched runs copy the entry out so
        # that one kept state does not pin its neighbours
        return slab[b] if slab.shape[0] == 1 else slab[b].copy()


def _lazy_binary(name: str) -> Any:
    def op(self: "LazyState", other: Any) -> Any:
        other = other._materialise() if isinstance(other, LazyState) else other
        return getattr(self._materialise(), name)(other)

    op.__name__ = name
    return op


class LazyState:
    """A state of ``results.states`` that is still a device snapshot; it turns into a ``QState`` (host) the first time
    anything but its shape / kind is asked of it.  Reads like the ``QState`` it becomes: ``np.asarray``, indexing,
    arithmetic and the ``Qobj``-style accessors all work on the materialised state."""

    __array_priority__ = 20.0

    def __init__(self, store: SnapshotStore, i: int, b: int, shape: tuple[int, int]) -> None:
        self._store = store
        self._i = int(i)
        self._b = int(b)
        self._shape = (int(shape[0]), int(shape[1]))
        self._q: QState | None = None

    def _materialise(self) -> QState:
        if self._q is None:
            self._q = QState(np.asarray(self._store.get(self._i, self._b)).reshape(self._shape))
            # a materialised state holds its host data and lets go of the store: keeping ONE state of a run alive must not
            # keep the 813 MB of the run's device snapshots alive with it
            self._store = None
        return self._q

    @property
    def shape(self) -> tuple[int, int]:
        return self._shape

    @property
    def ndim(self) -> int:
        return 2

    @property
    def dtype(self) -> Any:
        return np.dtype(complex)

    @property
    def isket(self) -> bool:
        return self._shape[1] == 1 and self._shape[0] > 1 or self._shape == (1, 1)

    @property
    def isoper(self) -> bool:
        return not self.isket

    @property
    def device_tensor(self) -> Any:
        """The snapshot on the GPU (a view), or None once this state (or the whole store) has moved to the host."""
        dev = None if self._store is None else self._store.device_tensor
        return None if dev is None else dev[self._i, self._b]

    def __array__(self, dtype: Any = None, copy: Any = None) -> np.ndarray:
        a = np.asarray(self._materialise())
        return a if dtype is None else a.astype(dtype)

    def __getattr__(self, name: str) -> Any:  # full, dag, diag, tr, norm, unit, overlap, copy, conj, T, real, ...
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._materialise(), name)

    def __getitem__(self, key: Any) -> Any:
        return self._materialise()[key]

    def __len__(self) -> int:
        return self._shape[0]

    def __iter__(self) -> Any:
        return iter(self._materialise())

    def __repr__(self) -> str:
        where = "host" if self._q is not None else "device snapshot"
        return f"LazyState(shape={self._shape}, {where})"

    def __neg__(self) -> Any:
        return -self._materialise()

    def __abs__(self) -> Any:
        return abs(self._materialise())

    def __eq__(self, other: Any) -> Any:  # elementwise, like the ndarray it stands for
        other = other._materialise() if isinstance(other, LazyState) else other
        return self._materialise() == other

    __hash__ = None  # type: ignore[assignment]

    for _n in ("add", "sub", "mul", "matmul", "truediv", "pow"):
        locals()[f"__{_n}__"] = _lazy_binary(f"__{_n}__")
        locals()[f"__r{_n}__"] = _lazy_binary(f"__r{_n}__")
    del _n


def multinomial(n_samples: int, probabilities: np.ndarray) -> np.ndarray:
    """pulser-core/pulser/math/multinomial.py:18-36."""
    rnd = np.random.rand(n_samples)
    cumsums = np.cumsum(probabilities)
    return np.searchsorted(cumsums, rnd)


class Result:
    """``pulser.result.Result`` (pulser-core/pulser/result.py:34-168)."""

    atom_order: tuple
    meas_basis: str

    @property
    def _size(self) -> int:
        return len(self.atom_order)

    def _weights(self) -> np.ndarray:  # pragma: no cover - abstract
        raise NotImplementedError

    @property
    def sampling_dist(self) -> dict[str, float]:
        """Sampling distribution of the measured bitstring (result.py:70-86)."""
        weights = self._weights()
        nonzero = np.argwhere(weights != 0.0).flatten()
        return {np.binary_repr(i, self._size): float(weights[i]) for i in nonzero}

    @property
    def sampling_errors(self) -> dict[str, float]:  # pragma: no cover - abstract
        raise NotImplementedError

    def get_samples(self, n_samples: int) -> Counter:
        """result.py:103-115 - insertion order = order of first occurrence."""
        return Counter(
            np.binary_repr(i, self._size)
            for i in multinomial(n_samples, self._weights())
        )

    def get_state(self) -> Any:
        raise NotImplementedError(f"`{self.__class__.__name__}.get_state()` is not implemented.")

    def plot_histogram(self, min_rate: float = 0.001, max_n_bitstrings: int | None = None,
                       show: bool = True) -> None:
        """Bar chart of the sampling distribution (result.py:128-153)."""
        import matplotlib.pyplot as plt

        probs = np.array(Counter(self.sampling_dist).most_common(max_n_bitstrings), dtype=object)
        probs = probs[probs[:, 1] >= min_rate]
        plt.bar(probs[:, 0], probs[:, 1])
        plt.xticks(rotation="vertical")
        plt.ylabel("Probabilites")
        if show:
            plt.show()

    def __str__(self) -> str:
        return self.__repr__()

    @classmethod
    def from_final_bitstrings(cls, atom_order: Sequence[str], total_duration: int,
                              final_bitstrings: Mapping[str, int]) -> "Result":
        raise NotImplementedError(f"'{cls.__name__}.from_final_bitstrings()' is not implemented.")


@dataclass
class SampledResult(Result):
    """``pulser.result.SampledResult`` (result.py:171-242)."""

    atom_order: tuple
    meas_basis: str
    bitstring_counts: Mapping[str, int]
    evaluation_time: float = 1.0

    def __post_init__(self) -> None:
        self.n_samples = sum(self.bitstring_counts.values())

    @property
    def final_bitstrings(self) -> Counter:
        """The measured bitstrings themselves (``Results.final_bitstrings``)."""
        return Counter(self.bitstring_counts)

    def get_samples(self, n_samples: int) -> Counter:
        warnings.warn(
            "'SampledResult.get_samples()' resamples a sampling distribution"
            " derived from the original 'bitstring_counts'. To get the real "
            "samples, accessing 'SampledResult.final_bitstrings' is "
            "recommended.",
            stacklevel=2,
        )
        return super().get_samples(n_samples)

    @property
    def sampling_errors(self) -> dict[str, float]:
        """Standard error of the mean of every bitstring's rate (result.py:204-213)."""
        return {
            bitstr: float(np.sqrt(p * (1 - p) / self.n_samples))
            for bitstr, p in self.sampling_dist.items()
        }

    def _weights(self) -> np.ndarray:
        weights = np.zeros(2**self._size)
        for bitstr, counts in self.bitstring_counts.items():
            weights[int(bitstr, base=2)] = counts / self.n_samples
        return weights / sum(weights)


@dataclass
class StateResult(Result):
    """``pulser_simulation.qutip_result.QutipResult`` with a NumPy state."""

    atom_order: tuple
    meas_basis: str
    state: QState
    matching_meas_basis: bool
    evaluation_time: float = 1.0

    @property
    def _dim(self) -> int:
        full = self.state.shape[0]
        return int(np.rint(full ** (1 / self._size)).astype(int))

    @property
    def sampling_errors(self) -> dict[str, float]:
        """An exact state has no sampling error (qutip_result.py:49-55)."""
        return {bitstr: 0.0 for bitstr in self.sampling_dist}

    @property
    def _basis_name(self) -> str:  # qutip_result.py:66-90
        if self.meas_basis == "XY":
            if self._dim == 3:
                return "XY_with_error"
            assert self._dim == 2, f"In XY, state's dimension can only be 2 or 3, not {self._dim}."
            return "XY"
        if self._dim == 4:
            return "all_with_error"
        if self._dim == 3:
            return self.meas_basis + "_with_error" if self.matching_meas_basis else "all"
        assert self._dim == 2, f"In Ising, state's dimension can be 2, 3 or 4, not {self._dim}."
        if not self.matching_meas_basis:
            return "digital" if self.meas_basis == "ground-rydberg" else "ground-rydberg"
        return self.meas_basis

    @property
    def _eigenbasis(self) -> list[str]:
        bases = self._basis_name.split("_with_error")
        names = ["ground-rydberg", "digital"] if bases[0] == "all" else [bases[0]]
        rank = ["u", "d", "r", "g", "h", "x"]
        states = {s for b in names for s in EIGENSTATES[b]}
        out = [s for s in rank if s in states]
        return out + (["x"] if len(bases) == 2 else [])

    def _weights(self) -> np.ndarray:
        """qutip_result.py:101-158."""
        size = self._size
        if not self.state.isket:
            probs = np.abs(self.state.diag())
        else:
            probs = (np.abs(self.state.full()) ** 2).flatten()
        if self._dim == 2:
            if self.matching_meas_basis:
                weights = probs[::-1] if self.meas_basis == "ground-rydberg" else probs
            else:
                weights = np.zeros(probs.size)
                weights[0] = 1.0
        elif self._dim in (3, 4):
            if self.meas_basis not in _ONE_STATE:
                raise RuntimeError(f"Unknown measurement basis '{self.meas_basis}'.")
            one = self._eigenbasis.index(_ONE_STATE[self.meas_basis])
            ex_one = [i for i in range(self._dim) if i != one]
            probs = probs.reshape([self._dim] * size)
            weights = np.zeros(2**size)
            for dec_val in range(2**size):
                ind = [ex_one if v == "0" else [one] for v in np.binary_repr(dec_val, width=size)]
                weights[dec_val] = np.sum(probs[np.ix_(*ind)])
        else:
            raise NotImplementedError(
                "Cannot sample system with single-atom state vectors of dimension > 4."
            )
        # builtin sum() = sequential left-to-right fp64 accumulation; the last
        # element of cumsum is the same sequence of additions, in C
        return weights / np.cumsum(weights)[-1]

    def get_state(
        self,
        reduce_to_basis: str | None = None,
        ignore_global_phase: bool = True,
        tol: float = 1e-6,
        normalize: bool = True,
    ) -> QState:
        """qutip_result.py:160-242: optional global-phase removal and reduction of a
        multi-level ket to one of its two-level bases (kets only, like the reference)."""
        state = QState(self.state.copy())
        if ignore_global_phase and state.isket:
            full = state.full()
            global_ph = float(np.angle(full[np.argmax(np.abs(full))])[0])
            state = QState(full * np.exp(-1j * global_ph))
        if self._dim == 2:
            if reduce_to_basis not in [None, self._basis_name]:
                raise TypeError(
                    f"Can't reduce a system in {self._basis_name}"
                    + f" to the {reduce_to_basis} basis."
                )
        elif reduce_to_basis is not None:
            if not state.isket:
                raise NotImplementedError(
                    "Reduce to basis not implemented for density matrix states."
                )
            if reduce_to_basis not in EIGENSTATES:
                raise ValueError(
                    "'reduce_to_basis' must be 'ground-rydberg', "
                    f"'XY', or 'digital', not '{reduce_to_basis}'."
                )
            eigenbasis = self._eigenbasis
            target = set(EIGENSTATES[reduce_to_basis])
            if not target.issubset(eigenbasis):
                raise ValueError(
                    f"Can't reduce a state expressed in {self._basis_name}"
                    f" into {reduce_to_basis}"
                )
            # basis states with every qudit inside the target basis survive
            d, n = self._dim, self._size
            kept_level = np.array([lvl in target for lvl in eigenbasis])
            digits = (np.arange(d**n)[:, None] // d ** np.arange(n - 1, -1, -1)[None, :]) % d
            keep = kept_level[digits].all(axis=1)
            amps = state.full()
            if not np.all(np.isclose(np.abs(amps[~keep]) ** 2, 0, atol=tol)):
                raise TypeError(
                    "Can't reduce to chosen basis because the population of a "
                    "state to eliminate is above the allowed tolerance."
                )
            reduced = amps[keep]
            if normalize:
                reduced = reduced / np.linalg.norm(reduced)
            state = QState(reduced)
        arr = np.asarray(state).copy()
        arr[np.abs(arr) < 1e-12] = 0  # Qobj.tidyup default atol
        return QState(arr)


class SimulationResults(collections.abc.Sequence):
    """simresults.py:37-229."""

    _use_pseudo_dens: bool = False

    def __init__(self, size: int, basis_name: str, sim_times: np.ndarray) -> None:
        self._size = size
        bases = ["ground-rydberg", "digital", "all", "XY"]
        bases += [b + "_with_error" for b in bases]
        if basis_name not in bases:
            raise ValueError(f"`basis_name` must be in {bases}")
        self._basis_name = basis_name
        self._dim = 3 if self._basis_name == "all" else 2
        if "_with_error" in self._basis_name:
            self._dim += 1
        self._sim_times = sim_times
        self._results_seq: tuple = ()

    def __getitem__(self, i: Any) -> Any:
        return self._results_seq[i]

    def __len__(self) -> int:
        return len(self._results_seq)

    @property
    def states(self) -> list:
        raise NotImplementedError

    @staticmethod
    def _as_operator(obs: Any) -> Any:
        """An observable as a dense array or a SciPy CSR matrix.  The reference takes ``qutip.Qobj`` or array-likes
        (simresults.py:110-118); a ``Qobj`` is recognised by its interface (``shape`` + ``full``) without importing qutip,
        and keeps its sparse storage when it offers one - a 14-atom occupation operator is 16 384 numbers, not 4.3 GB."""
        import scipy.sparse as sp

        if isinstance(obs, np.ndarray):
            return np.asarray(obs)
        if sp.issparse(obs):
            return obs.tocsr()
        if hasattr(obs, "full") and hasattr(obs, "shape"):
            for getter in (lambda: obs.data_as("csr_matrix"), lambda: obs.to("CSR").data.as_scipy(),
                           lambda: obs.data.as_scipy()):
                try:
                    m = getter()
                    if sp.issparse(m):
                        return m.tocsr()
                except Exception:  # another storage / another qutip version: the dense form below always exists
                    pass
            return np.asarray(obs.full())
        raise TypeError(
            f"Incompatible type {type(obs)} of observable. "
            "Type must be ArrayLike or qutip.Qobj."
        )

    def _expect_diagonal_on_device(self, states: Sequence[Any], diag: np.ndarray) -> np.ndarray | None:
        """<psi_t| diag(d) |psi_t> for every evaluation time straight from the device snapshots (kets of one run, none of
        them read yet): |psi|^2 . d as matrix-vector products over blocks of times - nothing crosses PCIe but the values.
        None when the states are not (all) device snapshots of one store."""
        live = [st for st in states if isinstance(st, LazyState) and st._store is not None and st._store.device_tensor is not None]
        if not live:
            return None
        store, b = live[0]._store, live[0]._b
        dev = store.device_tensor
        if not hasattr(dev, "real") or getattr(dev, "dim", lambda: 0)() != 3:  # [times, sequences, amplitudes]: kets only
            return None
        import torch

        on_dev = [isinstance(st, LazyState) and st._store is store and st._b == b for st in states]
        w = torch.from_numpy(np.ascontiguousarray(diag)).to(dev.device)
        rows = torch.as_tensor([st._i for st, f in zip(states, on_dev) if f], device=dev.device)
        vals = torch.empty(len(rows), dtype=w.dtype, device=dev.device)
        for c0 in range(0, len(rows), 128):
            x = dev[rows[c0:c0 + 128], b]
            p2 = x.real * x.real + x.imag * x.imag
            vals[c0:c0 + 128] = (p2.to(w.dtype) @ w)
        host = vals.cpu().numpy()
        out = np.empty(len(states), dtype=host.dtype)
        k = 0
        for i, (st, f) in enumerate(zip(states, on_dev)):
            if f:
                out[i] = host[k]
                k += 1
            else:  # the initial state, states that were read before (they live on the host), another run's states
                a = np.asarray(st).reshape(-1)
                out[i] = np.sum((a.real**2 + a.imag**2) * diag)
        return out

    def expect(self, obs_list: Sequence[Any]) -> list:
        """simresults.py:89-132 (``qutip.expect`` of the observables over the stored states).  Observables: arrays,
        ``qutip.Qobj`` (by interface) or SciPy sparse matrices; sparse ones stay sparse, and DIAGONAL ones (occupations,
        projectors, correlation products - what a Rydberg user plots) are evaluated on the device from the run's
        snapshots without reading a single state back."""
        import scipy.sparse as sp

        if not isinstance(obs_list, (list, np.ndarray)):
            raise TypeError("`obs_list` must be a list of operators.")
        dim = self._dim if not self._use_pseudo_dens else 2
        legal_shape = (dim**self._size, dim**self._size)
        mats = []
        for obs in obs_list:
            m = self._as_operator(obs)
            if tuple(m.shape) != legal_shape:
                raise ValueError(
                    "Incompatible shape of observable."
                    + f"Expected {legal_shape}, got {tuple(m.shape)}."
                )
            off = (m - sp.diags(m.diagonal())).count_nonzero() if sp.issparse(m) else np.count_nonzero(m - np.diag(np.diagonal(m)))
            if self._use_pseudo_dens and off:
                raise ValueError(f"Observable {obs!r} is non-diagonal.")
            mats.append((m, off == 0))
        if self._use_pseudo_dens:
            states = [self._calc_pseudo_density(i) for i in range(len(self))]
        else:
            states = self.states
        out = []
        for m, is_diag in mats:
            if sp.issparse(m):  # np.allclose(m, m^dag) for the stored entries: |m - m^dag| <= 1e-8 + 1e-5 |m^dag|
                dm = abs(m - m.conj().T).tocoo()
                ref = abs(m.conj().T).tocsr()
                herm = bool(np.all(dm.data <= 1e-8 + 1e-5 * np.asarray(ref[dm.row, dm.col]).reshape(-1))) if dm.nnz else True
            else:
                herm = bool(np.allclose(m, m.conj().T))
            if is_diag and not self._use_pseudo_dens:
                d = np.asarray(m.diagonal())
                fast = self._expect_diagonal_on_device(states, d.real.copy() if herm else d.astype(complex))
                if fast is not None:
                    out.append(np.array(fast.real if herm else fast))
                    continue
            vals = []
            for st in states:
                a = np.asarray(st)
                v = np.vdot(a, m @ a) if a.shape[1] == 1 and a.shape[0] > 1 else (m @ a).trace()
                vals.append(v.real if herm else v)
            out.append(np.array(vals))
        return out

    def to_host(self) -> "SimulationResults":
        """Move the stored states of this run from HBM to host memory now (what the reference always does,
        simulation.py:744-748).  ``results.states`` reads the same afterwards; the device snapshots are released."""
        seen: set[int] = set()
        for r in self._results_seq:
            st = getattr(r, "state", None)
            store = getattr(st, "_store", None)
            if store is not None and id(store) not in seen:
                seen.add(id(store))
                store.fetch_all()
        return self

    def sample_state(self, t: float, n_samples: int = 1000, t_tol: float = 1.0e-3) -> Counter:
        t_index = self._get_index_from_time(t, t_tol)
        return self[t_index].get_samples(n_samples)

    def sample_final_state(self, N_samples: int = 1000) -> Counter:
        return self.sample_state(self._sim_times[-1], N_samples)

    def plot(self, op: np.ndarray, fmt: str = "", label: str = "") -> None:
        """Expectation value of ``op`` over the evaluation times (simresults.py:164-174)."""
        import matplotlib.pyplot as plt

        plt.plot(self._sim_times, self.expect([op])[0], fmt, label=label)
        plt.xlabel("Time (µs)")
        plt.ylabel("Expectation value")

    def _get_index_from_time(self, t_float: float, tol: float = 1.0e-3) -> int:
        """simresults.py:176-190 - the FIRST index within tol."""
        try:
            return int(np.where(abs(t_float - self._sim_times) < tol)[0][0])
        except IndexError:
            raise IndexError(
                f"Given time {t_float} is absent from simulation times within"
                + f" tolerance {tol}."
            )

    def _meas_projector(self, state_n: int) -> np.ndarray:
        p = np.zeros((2, 2))
        good = 1 - state_n if self._basis_name == "ground-rydberg" else state_n
        p[good, good] = 1.0
        return p

    @lru_cache(maxsize=None)
    def _calc_pseudo_density(self, t_index: int) -> QState:
        """simresults.py:192-217: diagonal matrix of post-measurement weights."""
        w = self[t_index]._weights()
        D = 2**self._size
        diag = np.zeros(D)
        for i in np.nonzero(w)[0]:
            bits = np.binary_repr(i, width=self._size)
            d = np.array([1.0])
            for c in bits:
                d = np.kron(d, np.diag(self._meas_projector(int(c))))
            diag += w[i] * d
        return QState(np.diag(diag).astype(complex))


class NoisyResults(SimulationResults):
    """simresults.py:232-360."""

    _use_pseudo_dens = True

    def __init__(self, run_output: Sequence[SampledResult], size: int, basis_name: str,
                 sim_times: np.ndarray, n_measures: int) -> None:
        basis = basis_name.replace("_with_error", "")
        super().__init__(size, "digital" if basis == "all" else basis, sim_times)
        self.n_measures = n_measures
        self._results_seq = tuple(run_output)

    @property
    def states(self) -> list:
        return [self.get_state(t) for t in self._sim_times]

    @property
    def results(self) -> list[Counter]:
        return [Counter(res.sampling_dist) for res in self]

    def get_state(self, t: float, t_tol: float = 1.0e-3) -> QState:
        return self._calc_pseudo_density(self._get_index_from_time(t, t_tol))

    def get_final_state(self) -> QState:
        return self.get_state(self._sim_times[-1])

    def plot(self, op: np.ndarray, fmt: str = ".", label: str = "",  # type: ignore[override]
             error_bars: bool = True) -> None:
        """simresults.py:325-360: a diagonal observable with the standard error
        of the mean over ``n_measures`` shots as error bars."""
        import matplotlib.pyplot as plt

        if not error_bars:
            super().plot(op, fmt, label)
            return
        moy = self.expect([op])[0]
        sq = self.expect([np.asarray(op) @ np.asarray(op)])[0]
        st = np.sqrt(np.maximum(np.real(sq) - np.real(moy) ** 2, 0.0) / self.n_measures)
        plt.errorbar(self._sim_times, moy, st, fmt=fmt, lw=1, capsize=3, label=label)
        plt.xlabel("Time (µs)")
        plt.ylabel("Expectation value")


class CoherentResults(SimulationResults):
    """simresults.py:363-568."""

    def __init__(self, run_output: Sequence[StateResult], size: int, basis_name: str,
                 sim_times: np.ndarray, meas_basis: str,
                 meas_errors: Optional[Mapping[str, float]] = None) -> None:
        super().__init__(size, basis_name, sim_times)
        self._meas_basis = self._checked_meas_basis(meas_basis)
        self._results_seq = tuple(run_output)
        self._meas_errors = self._checked_meas_errors(meas_errors)
        if self._meas_errors is not None:  # sampled states become pseudo-densities (simresults.py:192-217)
            self._use_pseudo_dens = True

    def _checked_meas_basis(self, meas_basis: str) -> str:
        """Which measurement bases a simulation basis admits (contract of simresults.py:400-412)."""
        plain = self._basis_name.replace("_with_error", "")
        if "all" in self._basis_name:
            if meas_basis in ("ground-rydberg", "digital"):
                return meas_basis
            raise ValueError("`meas_basis` must be 'ground-rydberg' or 'digital'.")
        if meas_basis == plain:
            return meas_basis
        raise ValueError(f"`meas_basis` associated to basis_name '{self._basis_name}' must be '{plain}'.")

    @staticmethod
    def _checked_meas_errors(meas_errors: Optional[Mapping[str, float]]) -> Optional[Mapping[str, float]]:
        if meas_errors is None or sorted(meas_errors) == ["epsilon", "epsilon_prime"]:
            return meas_errors
        raise ValueError("When defining measurement errors, only values of 'epsilon' and 'epsilon_prime' "
                         "must be given.")

    @property
    def states(self) -> list[QState]:
        return [res.state for res in self]

    def get_state(self, t: float, reduce_to_basis: Optional[str] = None,
                  ignore_global_phase: bool = True, tol: float = 1e-6,
                  normalize: bool = True, t_tol: float = 1.0e-3) -> QState:
        t_index = self._get_index_from_time(t, t_tol)
        return self[t_index].get_state(reduce_to_basis, ignore_global_phase, tol, normalize)

    def get_final_state(self, reduce_to_basis: Optional[str] = None,
                        ignore_global_phase: bool = True, tol: float = 1e-6,
                        normalize: bool = True) -> QState:
        return self.get_state(self._sim_times[-1], reduce_to_basis, ignore_global_phase,
                              tol, normalize)

    def _meas_projector(self, state_n: int) -> np.ndarray:
        if self._meas_errors:  # simresults.py:500-520
            err = self._meas_errors["epsilon"] if state_n == 0 else self._meas_errors["epsilon_prime"]
            good = 1 - state_n if "ground-rydberg" in self._basis_name else state_n
            p = np.zeros((2, 2))
            p[good, good] = 1 - err
            p[1 - good, 1 - good] = err
            return p
        return super()._meas_projector(state_n)

    def sample_state(self, t: float, n_samples: int = 1000, t_tol: float = 1.0e-3) -> Counter:
        """simresults.py:522-568."""
        sampled_state = super().sample_state(t, n_samples, t_tol)
        if self._meas_errors is None or (
            self._meas_errors["epsilon"] == 0.0 and self._meas_errors["epsilon_prime"] == 0
        ):
            return sampled_state
        return spam_flips(sampled_state, self._meas_errors["epsilon"],
                          self._meas_errors["epsilon_prime"])


def spam_flips(sampled_state: Counter, eps: float, eps_p: float) -> Counter:
    """Measurement errors on already sampled bitstrings (behaviour of simresults.py:537-568).

    Every shot flips each of its bits independently: a measured 0 becomes 1 with
    probability ``eps`` (false positive), a measured 1 becomes 0 with ``eps_p``.  To
    reproduce the reference's Counters for a seed, the uniforms are drawn in ONE call of
    shape (total shots, n bits), rows ordered by the Counter's key order with each key
    repeated ``count`` times, and the result keeps first-occurrence order.  The
    bookkeeping is done on integer codes rather than on character arrays.
    """
    keys = list(sampled_state)
    if not keys:
        return Counter()
    width = len(keys[0])
    place = np.left_shift(1, np.arange(width - 1, -1, -1, dtype=np.int64))  # MSB first
    multiplicity = np.fromiter((sampled_state[k] for k in keys), dtype=np.int64, count=len(keys))
    codes = np.repeat(np.fromiter((int(k, 2) for k in keys), dtype=np.int64, count=len(keys)), multiplicity)
    ones = (codes[:, None] & place[None, :]) != 0
    u = np.random.uniform(size=(int(multiplicity.sum()), width))
    toggled = u < np.where(ones, eps_p, eps)
    after = codes ^ (toggled * place[None, :]).sum(axis=1)
    values, first_seen, counts = np.unique(after, return_index=True, return_counts=True)
    order = np.argsort(first_seen, kind="stable")
    return Counter({np.binary_repr(int(values[i]), width): int(counts[i]) for i in order})
