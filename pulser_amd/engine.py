"""Python face of the C ABI: device tables in, torch tensors as state storage.

PyTorch is plumbing only (device memory, streams).  All arithmetic of the hot
path happens in the hand-written HIP kernels of ``csrc/rydemu.hip`` through
``librydemu.so``; this module never computes a fallback on the host.
"""

from __future__ import annotations

import ctypes as C
from typing import Any, Mapping, Sequence

import numpy as np

from . import _lib
from ._lib import RYD_MESOLVE, RYD_SESOLVE, RydConfig, RydOpts, RydQDesc, RydStats
from .terms import DeviceTables, lower


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "pulser_amd needs an AMD GPU visible to PyTorch-ROCm "
            "(torch.cuda.is_available() is False); there is no CPU fallback."
        )
    return torch


# ryd_opts.method: propagator of the multi-launch sesolve path ("auto": split-operator for
# two-level kets of 15+ atoms, Taylor polynomial otherwise)
_METHODS = {"auto": 0, "krylov": 1, "split": 2, "taylor": 3}


def _method_code(method: str) -> int:
    try:
        return _METHODS[method]
    except KeyError:
        raise ValueError(f"unknown method {method!r}; one of {sorted(_METHODS)}") from None


def outer_accumulate(psi: Any, acc: Any, weights: Any = None) -> None:
    """``acc[D, D] += sum_b w_b |psi_b><psi_b|`` on the device, without a solver handle and for any
    state dimension (``ryd_outer_accumulate_dim``): the trajectory mean of
    pulser_simulation/aggregators.py:20-37.  ``psi`` complex128[B, D] and ``acc`` complex128[D, D]
    are contiguous CUDA tensors on the same device."""
    torch = _torch()
    if not (psi.is_cuda and acc.is_cuda and psi.device == acc.device):
        raise ValueError("psi and acc must be CUDA tensors on the same device")
    if psi.dtype != torch.complex128 or acc.dtype != torch.complex128:
        raise TypeError("complex128 tensors expected")
    if psi.dim() != 2 or tuple(acc.shape) != (psi.shape[1], psi.shape[1]):
        raise ValueError(f"shapes {tuple(psi.shape)} / {tuple(acc.shape)} do not match [B, D] / [D, D]")
    if not (psi.is_contiguous() and acc.is_contiguous()):
        raise ValueError("contiguous tensors expected")
    wptr = None
    if weights is not None:
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        if weights.shape != (psi.shape[0],):
            raise ValueError(f"need one weight per state ({psi.shape[0]}), got {weights.shape}")
        wptr = weights.ctypes.data
    _lib.check(_lib.load().ryd_outer_accumulate_dim(
        psi.data_ptr(), int(psi.shape[0]), int(psi.shape[1]), wptr, acc.data_ptr(), int(psi.device.index or 0),
        torch.cuda.current_stream(psi.device).cuda_stream))


def accumulate(x: Any, acc: Any, weight: float = 1.0) -> None:
    """``acc += weight * x`` elementwise on the device (``ryd_accumulate``): the running sum of
    trajectory density matrices of aggregators.py:29-35."""
    torch = _torch()
    if not (x.is_cuda and acc.is_cuda and x.device == acc.device) or x.shape != acc.shape:
        raise ValueError("x and acc must be CUDA tensors of one shape on the same device")
    if x.dtype != torch.complex128 or acc.dtype != torch.complex128 or not (x.is_contiguous() and acc.is_contiguous()):
        raise TypeError("contiguous complex128 tensors expected")
    _lib.check(_lib.load().ryd_accumulate(
        x.data_ptr(), float(weight), int(x.numel()), acc.data_ptr(), int(x.device.index or 0),
        torch.cuda.current_stream(x.device).cuda_stream))


class Engine:
    """One ``ryd_handle``: a batch of B states of N atoms on one device.

    Mirrors the role of the ``QobjEvo`` + solver pair the reference builds at
    pulser-simulation/pulser_simulation/simulation.py:729-735.
    """

    def __init__(
        self,
        tables: DeviceTables,
        mode: str = "sesolve",
        device: int | None = None,
        tile_bits: int = 0,
    ) -> None:
        self.torch = _torch()
        self.lib = _lib.load()
        self.tables = tables
        self.n = tables.n_qubits
        self.batch = tables.batch
        if mode not in ("sesolve", "mesolve", "mcsolve"):
            raise ValueError(f"unknown mode {mode!r}")
        self.mode = RYD_MESOLVE if mode == "mesolve" else RYD_SESOLVE
        self.monte_carlo = mode == "mcsolve"
        self.device_index = (
            self.torch.cuda.current_device() if device is None else int(device)
        )
        self.device = self.torch.device("cuda", self.device_index)
        cfg = RydConfig(
            abi_version=_lib.RYD_ABI_VERSION,
            n_qubits=self.n,
            batch=self.batch,
            mode=self.mode,
            device=self.device_index,
            tile_bits=tile_bits,
        )
        self._h = C.c_void_p()
        _lib.check(self.lib.ryd_create(C.byref(cfg), C.byref(self._h)))
        self._upload()

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ryd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self) -> "Engine":
        return self

    def __exit__(self, *exc: Any) -> None:
        self.close()

    # -- tables -----------------------------------------------------------
    def _upload(self) -> None:
        t = self.tables
        tk = np.ascontiguousarray(t.tknots, dtype=np.float64)
        pp = np.ascontiguousarray(t.pp, dtype=np.complex128)
        _lib.check(
            self.lib.ryd_set_series(
                self._h, pp.shape[0], len(tk), tk.ctypes.data, pp.ctypes.data
            )
        )
        desc = np.ascontiguousarray(t.desc)
        assert desc.dtype.itemsize == C.sizeof(RydQDesc)
        _lib.check(self.lib.ryd_set_qubit_desc(self._h, desc.ctypes.data))
        if t.dterms is not None and len(t.dterms):
            dt = np.ascontiguousarray(t.dterms)
            assert dt.dtype.itemsize == C.sizeof(_lib.RydDTerm)
            _lib.check(self.lib.ryd_set_detuning_terms(self._h, len(dt), dt.ctypes.data))
        u = np.ascontiguousarray(t.interaction, dtype=np.float64)
        _lib.check(self.lib.ryd_set_interaction(self._h, u.ctypes.data, u.shape[0]))
        if self.mode == RYD_MESOLVE and t.dissipator is not None:
            s = np.ascontiguousarray(t.dissipator, dtype=np.complex128)
            _lib.check(self.lib.ryd_set_dissipator(self._h, s.ctypes.data))
        if self.monte_carlo:
            if t.collapse_local is None:
                raise ValueError("mcsolve needs collapse operators (tables.collapse_local)")
            c = np.ascontiguousarray(t.collapse_local, dtype=np.complex128)
            if c.shape[1:] != (2, 2):
                raise NotImplementedError("Monte-Carlo trajectories need 2-level collapse operators")
            _lib.check(self.lib.ryd_set_collapse(self._h, c.shape[0], c.ctypes.data))

    @classmethod
    def from_problems(
        cls, problems: Sequence[Mapping[str, Any]], mode: str | None = None, **kw: Any
    ) -> "Engine":
        tables = lower(problems)
        if mode is None:
            mode = "mesolve" if tables.dissipator is not None else "sesolve"
        return cls(tables, mode=mode, **kw)

    # -- state helpers ----------------------------------------------------
    @property
    def dim(self) -> int:
        return 1 << self.n

    @property
    def state_shape(self) -> tuple[int, ...]:
        if self.mode == RYD_MESOLVE:
            return (self.batch, self.dim, self.dim)
        return (self.batch, self.dim)

    def _stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _check_state(self, x: Any, shape: tuple[int, ...] | None = None) -> None:
        torch = self.torch
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.is_contiguous()):
            raise TypeError("state must be a contiguous CUDA/HIP torch tensor")
        if x.dtype != torch.complex128:
            raise TypeError("state must be complex128")
        if tuple(x.shape) != (shape or self.state_shape):
            raise ValueError(
                f"Incompatible shape of state. Expected {shape or self.state_shape}, "
                f"got {tuple(x.shape)}."
            )

    def new_state(self, kets: np.ndarray | None = None) -> Any:
        """Device state from host ket(s) ``complex[B?, 2^N]``.

        Default (``kets=None``): basis vector ``2^N - 1``, every atom in local state 1.
        That is ``|g...g>`` in the ground-rydberg order ``[r, g]`` (simulation.py:498-505)
        - and only there: in the digital order ``[g, h]`` it would be ``|h...h>``, so the
        front-end (``QutipEmulator``) always passes the initial ket explicitly."""
        torch = self.torch
        if kets is None:
            psi = torch.zeros((self.batch, self.dim), dtype=torch.complex128, device=self.device)
            psi[:, -1] = 1.0
        else:
            host = np.asarray(kets, dtype=np.complex128).reshape(-1, self.dim)
            if host.shape[0] == 1 and self.batch > 1:
                # one ket for the whole batch: upload it once, replicate on the device (a batch of 256
                # 12-atom kets is 16 MB of pageable host memory otherwise: 9 ms per block of trajectories)
                one = torch.from_numpy(np.ascontiguousarray(host)).to(self.device)
                psi = one.expand(self.batch, self.dim).contiguous()
            else:
                psi = torch.from_numpy(np.ascontiguousarray(host)).to(self.device)
        if self.mode != RYD_MESOLVE:
            return psi
        rho = torch.empty(self.state_shape, dtype=torch.complex128, device=self.device)
        _lib.check(
            self.lib.ryd_ket_to_dm(self._h, psi.data_ptr(), rho.data_ptr(), self._stream())
        )
        return rho

    # -- hot path ---------------------------------------------------------
    def evolve(
        self,
        state: Any,
        t0: float,
        t1: float,
        taylor_order: int = 0,
        tol: float = 0.0,
        max_step: float = 0.0,
        max_order: int = 0,
        magnus_tol: float = 0.0,
        split_steps: int = 0,
        method: str = "auto",
    ) -> None:
        """In place: ``state <- U(t1, t0) state`` (times in us)."""
        self._check_state(state)
        opts = RydOpts(
            taylor_order=int(taylor_order),
            max_order=int(max_order),
            tol=float(tol),
            max_step=float(max_step),
            magnus_tol=float(magnus_tol),
            split_steps=int(split_steps),
            method=_method_code(method),
        )
        _lib.check(
            self.lib.ryd_evolve(
                self._h, state.data_ptr(), float(t0), float(t1), C.byref(opts), self._stream()
            )
        )
        self._split_budget_check(method, float(tol))

    def _split_budget_check(self, method: str, tol: float) -> None:
        """The split-operator controller books its local-error estimates in ``ryd_stats.reserved[0]``;
        when it could not hold the sequence budget (retries used up, nothing to roll back to) say so."""
        if not (method == "split" or (method == "auto" and self.n >= 12)):
            return  # (12 - 14 atoms may, 15+ atoms do take the split-operator path by default; quantum jumps included:
            #          a jump solve cannot roll back, so an overrun is only ever booked - and must be reported; the
            #          split-operator master equation of 12 - 14 atoms books its a-priori estimate the same way)
        est = self.stats()["reserved"][0]
        budget = 500.0 * tol if tol > 0 else 8e-8  # (host_split.hpp: kSplitTolTotal, a bound on the 2-norm of the error since round 6; the bar itself is 1e-7)
        if est > (1000.0 * tol if tol > 0 else 1e-7):
            import warnings

            warnings.warn(
                f"split-operator step-size controller: accumulated local-error estimate {est:.1e} exceeds "
                f"the budget {budget:.1e} of this solve; tighten `tol` or use method='taylor'.",
                RuntimeWarning, stacklevel=3)

    def solve(
        self,
        state: Any,
        times: Sequence[float],
        store: bool = True,
        taylor_order: int = 0,
        tol: float = 0.0,
        max_step: float = 0.0,
        max_order: int = 0,
        magnus_tol: float = 0.0,
        split_steps: int = 0,
        method: str = "auto",
    ) -> Any:
        """Advance ``state`` in place through ``times`` (us); with ``store``
        return complex128[len(times)-1, B, dim...] = the states at times[1:]
        (``result.states[1:]`` of the reference's solver call,
        simulation.py:729-748)."""
        self._check_state(state)
        t = np.ascontiguousarray(times, dtype=np.float64)
        if t.ndim != 1 or len(t) < 2:
            raise ValueError("times must hold at least two values")
        out = None
        if store:
            out = self.torch.empty(
                (len(t) - 1,) + self.state_shape,
                dtype=self.torch.complex128,
                device=self.device,
            )
        opts = RydOpts(
            taylor_order=int(taylor_order),
            max_order=int(max_order),
            tol=float(tol),
            max_step=float(max_step),
            magnus_tol=float(magnus_tol),
            split_steps=int(split_steps),
            method=_method_code(method),
        )
        _lib.check(
            self.lib.ryd_solve(
                self._h,
                state.data_ptr(),
                len(t),
                t.ctypes.data,
                out.data_ptr() if out is not None else None,
                C.byref(opts),
                self._stream(),
            )
        )
        self._split_budget_check(method, float(tol))
        return out

    def mc_solve(
        self,
        state: Any,
        times: Sequence[float],
        seeds: Sequence[int],
        store: bool = True,
        taylor_order: int = 0,
        tol: float = 0.0,
        max_step: float = 0.0,
        max_order: int = 0,
        magnus_tol: float = 0.0,
        method: str = "auto",
    ) -> Any:
        """One quantum-jump trajectory per batch entry (``qutip.mcsolve`` with
        ``ntraj=1``, simulation.py:705-735): ``seeds`` holds one uint64 per batch
        entry.  Returns the normalised kets at times[1:] like :meth:`solve`."""
        if not self.monte_carlo:
            raise RuntimeError("engine was not created with mode='mcsolve'")
        self._check_state(state)
        t = np.ascontiguousarray(times, dtype=np.float64)
        if t.ndim != 1 or len(t) < 2:
            raise ValueError("times must hold at least two values")
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        if sd.shape != (self.batch,):
            raise ValueError(f"need one seed per batch entry ({self.batch}), got {sd.shape}")
        out = None
        if store:
            out = self.torch.empty((len(t) - 1,) + self.state_shape, dtype=self.torch.complex128,
                                   device=self.device)
        opts = RydOpts(taylor_order=int(taylor_order), max_order=int(max_order), tol=float(tol),
                       max_step=float(max_step), magnus_tol=float(magnus_tol), method=_method_code(method))
        _lib.check(
            self.lib.ryd_mc_solve(
                self._h, state.data_ptr(), len(t), t.ctypes.data,
                out.data_ptr() if out is not None else None, sd.ctypes.data, C.byref(opts),
                self._stream(),
            )
        )
        self._split_budget_check(method, float(tol))
        return out

    def mc_jumps(self) -> np.ndarray:
        """Number of collapses of every trajectory of the last :meth:`mc_solve`."""
        counts = np.zeros(self.batch, dtype=np.int32)
        _lib.check(self.lib.ryd_mc_get_jumps(self._h, counts.ctypes.data, self._stream()))
        return counts

    def set_path(self, force_generic: bool, no_tile14: bool = False,
                 force_tile14: bool = False, no_single_pass: bool = False,
                 force_single_pass: bool = False, no_ket: bool = False,
                 force_ket: bool = False, no_split: bool = False,
                 split_fixed: bool = False, split_no_loop: bool = False,
                 no_merge: bool = False, split_small_tiles: bool = False, split_s6: bool = False,
                 no_split14: bool = False, split_turns: bool = False, rows_ket: bool = False,
                 snaps_outside: bool = False) -> None:
        """Test/bench hook: disable the persistent small-N kernel and/or the 2^14
        register-tile kernel with the Hermitian mesolve path (the tiled
        multi-pass kernels are used instead), or force the register tiles;
        ``no_single_pass`` disables the one-launch plan of states up to 128 MiB;
        ``no_ket`` disables the register-resident ket kernel and the split-operator
        master equation built on it, ``force_ket`` uses them from 10 atoms on;
        ``no_split`` keeps the Taylor polynomial for kets of 15+ atoms, ``split_fixed``
        switches the step-size control of the split-operator path off; ``no_merge`` keeps
        every CF4 step inside one knot interval (no multi-knot steps on smooth stretches);
        ``split_small_tiles`` keeps the 2^12 tiles of the split-operator passes where 2^14 tiles are the default
        (21 - 23 atoms); ``no_split14`` keeps batches of 14-atom sequences on the polynomial
        register-resident kernel (k_ket) instead of the split-operator one (k_split14_loop);
        ``rows_ket`` keeps the row passes of the split-operator master equation on k_ket (round 3) instead of k_split_reg;
        ``split_turns`` runs them on the round-3 kernel (two LDS turns per stage) instead of k_split_reg;
        ``snaps_outside``: every evaluation time closes a run of k_split_reg (round 4) instead of a snapshot stored from
        the registers inside the run;
        ``split_s6`` keeps the 4th-order composition with sub-steps that end at every knot
        (round 2) where the 6th-order one with multi-knot sub-steps is the default."""
        _lib.check(self.lib.ryd_set_path(
            self._h, int(bool(force_generic)) | (2 if no_single_pass else 0)
            | (4 if no_tile14 else 0) | (8 if force_tile14 else 0)
            | (16 if force_single_pass else 0) | (32 if no_ket else 0)
            | (64 if force_ket else 0) | (128 if no_split else 0)
            | (256 if split_fixed else 0) | (512 if split_no_loop else 0)
            | (1024 if no_merge else 0) | (2048 if split_small_tiles else 0) | (8192 if split_s6 else 0)
            | (16384 if no_split14 else 0) | (32768 if split_turns else 0) | (65536 if rows_ket else 0)
            | (131072 if snaps_outside else 0)))

    def apply_generator(self, x: Any, t: float) -> Any:
        """``G(t) x`` with ``G = -iH`` (sesolve) or the Lindbladian (mesolve)."""
        self._check_state(x)
        out = self.torch.empty_like(x)
        _lib.check(
            self.lib.ryd_apply_generator(
                self._h, x.data_ptr(), out.data_ptr(), float(t), self._stream()
            )
        )
        return out

    def probabilities(self, state: Any, reverse: bool = False) -> Any:
        self._check_state(state)
        w = self.torch.empty((self.batch, self.dim), dtype=self.torch.float64, device=self.device)
        _lib.check(
            self.lib.ryd_probabilities(
                self._h, state.data_ptr(), w.data_ptr(), int(bool(reverse)), self._stream()
            )
        )
        return w

    def observe(self, state: Any, t: float, occupation: bool = True, correlation: bool = True,
                energy: bool = True, density: bool = False) -> dict[str, np.ndarray]:
        """``ryd_observe``: occupations, correlation matrix and energy moments of every batch
        entry in one device call (no per-observable launches, no H(t) materialisation).
        ``density``: ``state`` is a density matrix ``[B, D, D]`` observed with this (ket) engine's
        Hamiltonian.  Returns host arrays: ``norm2`` [B] (trace for density matrices),
        ``occupation`` [B, N], ``correlation`` [B, N, N], ``energy`` [B], ``energy2`` [B] - NOT
        normalised (divide by ``norm2``)."""
        if density and self.mode != RYD_MESOLVE:
            want = (self.batch, self.dim, self.dim)
            if (tuple(state.shape) != want or state.dtype != self.torch.complex128
                    or not state.is_contiguous() or state.device != self.device):
                raise ValueError(f"density matrix must be a contiguous complex128 tensor of shape {want} on {self.device}")
        else:
            self._check_state(state)
        n = self.n
        what = (1 if occupation else 0) | (2 if correlation else 0) | (4 if energy else 0) | (8 if density else 0)
        out = self.torch.empty((self.batch, n * n + n + 3), dtype=self.torch.float64, device=self.device)
        _lib.check(self.lib.ryd_observe(self._h, state.data_ptr(), float(t), what, out.data_ptr(),
                                        self._stream()))
        o = out.cpu().numpy()
        return {"occupation": o[:, :n], "norm2": o[:, n],
                "correlation": o[:, n + 1:n + 1 + n * n].reshape(self.batch, n, n),
                "energy": o[:, n * n + n + 1], "energy2": o[:, n * n + n + 2]}

    def occupations(self, state: Any) -> Any:
        """float64[B, N+1]: <n_k> and, last, the squared norm / trace."""
        self._check_state(state)
        out = self.torch.empty((self.batch, self.n + 1), dtype=self.torch.float64, device=self.device)
        _lib.check(
            self.lib.ryd_occupations(self._h, state.data_ptr(), out.data_ptr(), self._stream())
        )
        return out

    def outer_accumulate(self, psi: Any, acc: Any, weights: np.ndarray | None = None) -> None:
        self._check_state(psi, (self.batch, self.dim))
        self._check_state(acc, (self.dim, self.dim))
        wptr = None
        if weights is not None:
            weights = np.ascontiguousarray(weights, dtype=np.float64)
            assert weights.shape == (self.batch,)
            wptr = weights.ctypes.data
        _lib.check(
            self.lib.ryd_outer_accumulate(
                self._h, psi.data_ptr(), wptr, acc.data_ptr(), self._stream()
            )
        )

    # -- introspection ----------------------------------------------------
    def stats(self) -> dict[str, Any]:
        s = RydStats()
        _lib.check(self.lib.ryd_get_stats(self._h, C.byref(s)))
        return {
            "n_applications": int(s.n_applications),
            "n_launches": int(s.n_launches),
            "n_steps": int(s.n_steps),
            "passes": int(s.passes),
            "last_order": int(s.last_order),
            "norm_bound": float(s.norm_bound),
            # split-operator path: accumulated local-error estimate of the last solve, last
            # measured local error, its sub-step (us), checkpoint restores
            "reserved": [float(v) for v in s.reserved],
        }

    def reset_stats(self) -> None:
        _lib.check(self.lib.ryd_reset_stats(self._h))

    def set_kernel_timing(self, enable: bool) -> None:
        _lib.check(self.lib.ryd_set_kernel_timing(self._h, int(bool(enable))))

    def kernel_timing(self) -> tuple[float, int]:
        ms = C.c_double()
        n = C.c_int64()
        _lib.check(self.lib.ryd_get_kernel_timing(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)


class GeneralEngine:
    """Explicit-term engine for the systems the tuned kernels do not cover
    (multi-level bases, leakage, XY, arbitrary eff_noise); see
    ``pulser_amd.general``.  One problem per engine, batch = 1."""

    def __init__(self, tables: Any, device: int | None = None) -> None:
        from ._lib import RydGeneralConfig

        self.torch = _torch()
        self.lib = _lib.load()
        self.tables = tables
        self.dim = int(tables.dim)
        self.local_dim = int(tables.local_dim)
        self.n = int(tables.n_qudits)
        self.is_density = bool(tables.is_density)
        self.batch = 1
        self.device_index = self.torch.cuda.current_device() if device is None else int(device)
        self.device = self.torch.device("cuda", self.device_index)
        cfg = RydGeneralConfig(abi_version=_lib.RYD_ABI_VERSION, batch=1,
                               device=self.device_index, reserved=0, dim=self.dim)
        self._h = C.c_void_p()
        _lib.check(self.lib.ryd_general_create(C.byref(cfg), C.byref(self._h)))
        tk = np.ascontiguousarray(tables.tknots, dtype=np.float64)
        pp = np.ascontiguousarray(tables.pp, dtype=np.complex128)
        _lib.check(self.lib.ryd_set_series(self._h, pp.shape[0], len(tk), tk.ctypes.data,
                                           pp.ctypes.data))
        for i in range(len(tables.values)):
            free = tables.free[i] if getattr(tables, "free", None) is not None else None
            tail = (int(tables.series[i]), int(tables.conj[i]), float(tables.scale[i].real),
                    float(tables.scale[i].imag), float(tables.row_norm[i]))
            if free is not None and free[0] == "diag":
                v = np.ascontiguousarray(free[1], dtype=np.complex128)
                _lib.check(self.lib.ryd_general_add_diag_term(self._h, v.ctypes.data, *tail))
                continue
            if free is not None:
                _, d, p, st, w, rr, cc, vals = free
                st = np.ascontiguousarray(st, dtype=np.int64)
                w = np.ascontiguousarray(w, dtype=np.float64)
                rr = np.ascontiguousarray(rr, dtype=np.int32)
                cc = np.ascontiguousarray(cc, dtype=np.int32)
                vals = np.ascontiguousarray(vals, dtype=np.complex128)
                _lib.check(self.lib.ryd_general_add_local_term(
                    self._h, int(d), int(p), len(w), st.ctypes.data, w.ctypes.data, len(vals), rr.ctypes.data,
                    cc.ctypes.data, vals.ctypes.data, *tail))
                continue
            rp, ci, va = tables.row_ptr[i], tables.col_idx[i], tables.values[i]
            _lib.check(self.lib.ryd_general_add_term(
                self._h, len(va), rp.ctypes.data, ci.ctypes.data if len(va) else None,
                va.ctypes.data if len(va) else None, int(tables.series[i]), int(tables.conj[i]),
                float(tables.scale[i].real), float(tables.scale[i].imag), float(tables.row_norm[i])))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.ryd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self) -> "GeneralEngine":
        return self

    def __exit__(self, *exc: Any) -> None:
        self.close()

    def _stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def new_state(self, ket: np.ndarray) -> Any:
        """Device vector from a host ket: psi, or row-major vec(|psi><psi|)."""
        v = np.asarray(ket, dtype=np.complex128).reshape(-1)
        if self.is_density:
            v = np.outer(v, v.conj()).reshape(-1)
        if v.size != self.dim:
            raise ValueError(f"Incompatible shape of state. Expected {self.dim}, got {v.size}.")
        return self.torch.from_numpy(np.ascontiguousarray(v[None, :])).to(self.device)

    def solve(self, state: Any, times: Sequence[float], **opts: Any) -> Any:
        t = np.ascontiguousarray(times, dtype=np.float64)
        out = self.torch.empty((len(t) - 1, 1, self.dim), dtype=self.torch.complex128,
                               device=self.device)
        o = RydOpts(taylor_order=int(opts.get("taylor_order", 0)), max_order=int(opts.get("max_order", 0)),
                    tol=float(opts.get("tol", 0.0)), max_step=float(opts.get("max_step", 0.0)),
                    magnus_tol=float(opts.get("magnus_tol", 0.0)))
        _lib.check(self.lib.ryd_solve(self._h, state.data_ptr(), len(t), t.ctypes.data,
                                      out.data_ptr(), C.byref(o), self._stream()))
        return out

    @staticmethod
    def solve_many(engines: Sequence["GeneralEngine"], states: Sequence[Any], times: Sequence[float],
                   **opts: Any) -> list[Any]:
        """Advance ``states[b]`` (in place) of every engine through ``times`` in ONE launch
        (``ryd_general_solve_many``: one workgroup per problem, vectors of at most 4096 entries) and return
        the snapshots complex128[len(times) - 1, 1, dim_b] per engine - the noise trajectories of a
        multi-level run (simulation.py:903-915)."""
        if not engines:
            return []
        e0 = engines[0]
        t = np.ascontiguousarray(times, dtype=np.float64)
        outs = [e.torch.empty((len(t) - 1, 1, e.dim), dtype=e.torch.complex128, device=e.device) for e in engines]
        n = len(engines)
        hs = (C.c_void_p * n)(*[e._h for e in engines])
        sp = (C.c_void_p * n)(*[s.data_ptr() for s in states])
        op = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        o = RydOpts(taylor_order=int(opts.get("taylor_order", 0)), max_order=int(opts.get("max_order", 0)),
                    tol=float(opts.get("tol", 0.0)), max_step=float(opts.get("max_step", 0.0)),
                    magnus_tol=float(opts.get("magnus_tol", 0.0)))
        _lib.check(e0.lib.ryd_general_solve_many(hs, n, sp, len(t), t.ctypes.data, op, C.byref(o), e0._stream()))
        return outs

    def apply_generator(self, x: Any, t: float) -> Any:
        # the C side trusts the pointer: a mis-sized vector would read / write out of bounds
        if (tuple(x.shape) != (1, self.dim) or x.dtype != self.torch.complex128
                or not x.is_contiguous() or x.device != self.device):
            raise ValueError(f"state must be a contiguous complex128 tensor of shape (1, {self.dim}) "
                             f"on {self.device}, got {tuple(x.shape)} {x.dtype} on {x.device}")
        out = self.torch.empty_like(x)
        _lib.check(self.lib.ryd_apply_generator(self._h, x.data_ptr(), out.data_ptr(), float(t),
                                                self._stream()))
        return out

    def set_path(self, force_multi_launch: bool, no_sites: bool = False, no_fused: bool = False) -> None:
        """Test hook: one launch per Taylor stage instead of the persistent
        one-launch kernel used for vectors of at most 4096 entries; ``no_sites``: the term-by-term
        kernel instead of the site-fused application of matrix-free terms; ``no_fused``: the round-3 site kernel
        (k_gen_apply_sites) instead of the padded site tables of k_gen_apply_fused."""
        _lib.check(self.lib.ryd_set_path(self._h, int(bool(force_multi_launch)) | (4096 if no_sites else 0)
                                         | (262144 if no_fused else 0)))

    def stats(self) -> dict[str, Any]:
        s = RydStats()
        _lib.check(self.lib.ryd_get_stats(self._h, C.byref(s)))
        return {"n_applications": int(s.n_applications), "n_launches": int(s.n_launches),
                "n_steps": int(s.n_steps), "passes": 1, "last_order": int(s.last_order),
                "norm_bound": float(s.norm_bound)}
