"""ctypes binding of ``librydemu.so`` (C ABI in ``include/rydemu.h``).

The product has NO CPU fallback: if the HIP library is missing or does not
export the ABI, importing the engine fails loudly.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librydemu.so")

RYD_ABI_VERSION = 1
RYD_MAX_QUBITS = 30
RYD_SESOLVE, RYD_MESOLVE = 0, 1


class RydError(RuntimeError):
    """Error reported by librydemu (message from ``ryd_last_error``)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"librydemu error {code}: {msg}")
        self.code = code


class RydConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_qubits", C.c_int32),
        ("batch", C.c_int32),
        ("mode", C.c_int32),
        ("device", C.c_int32),
        ("tile_bits", C.c_int32),
        ("reserved", C.c_int32 * 2),
    ]


class RydGeneralConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("batch", C.c_int32),
        ("device", C.c_int32),
        ("reserved", C.c_int32),
        ("dim", C.c_int64),
    ]


class RydQDesc(C.Structure):
    _fields_ = [
        ("drive_series", C.c_int32),
        ("det_series", C.c_int32),
        ("off_series", C.c_int32),
        ("extra", C.c_int32),
        ("drive_scale", C.c_double),
        ("det_scale", C.c_double),
        ("off_scale", C.c_double),
    ]


class RydDTerm(C.Structure):
    _fields_ = [("series", C.c_int32), ("remaining", C.c_int32), ("scale", C.c_double)]


class RydOpts(C.Structure):
    _fields_ = [
        ("taylor_order", C.c_int32),
        ("max_order", C.c_int32),
        ("tol", C.c_double),
        ("max_step", C.c_double),
        ("magnus_tol", C.c_double),
        ("split_steps", C.c_int32),
        ("method", C.c_int32),
        ("reserved", C.c_double * 2),
    ]


class RydStats(C.Structure):
    _fields_ = [
        ("n_applications", C.c_int64),
        ("n_launches", C.c_int64),
        ("n_steps", C.c_int64),
        ("passes", C.c_int32),
        ("last_order", C.c_int32),
        ("norm_bound", C.c_double),
        ("reserved", C.c_double * 4),
    ]


# name -> (restype, argtypes); every symbol include/rydemu.h declares
SYMBOLS = {
    "ryd_create": (C.c_int, [C.POINTER(RydConfig), C.POINTER(C.c_void_p)]),
    "ryd_destroy": (None, [C.c_void_p]),
    "ryd_set_series": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ryd_set_qubit_desc": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ryd_set_detuning_terms": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ryd_set_interaction": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "ryd_set_dissipator": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ryd_evolve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.POINTER(RydOpts), C.c_void_p]),
    "ryd_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(RydOpts), C.c_void_p]),
    "ryd_set_collapse": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ryd_mc_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(RydOpts), C.c_void_p]),
    "ryd_mc_get_jumps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ryd_set_path": (C.c_int, [C.c_void_p, C.c_int32]),
    "ryd_general_create": (C.c_int, [C.POINTER(RydGeneralConfig), C.POINTER(C.c_void_p)]),
    "ryd_general_add_term": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]),
    "ryd_general_add_local_term": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                             C.c_double, C.c_double, C.c_double]),
    "ryd_general_add_diag_term": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                            C.c_double]),
    "ryd_general_solve_many": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.POINTER(RydOpts), C.c_void_p]),
    "ryd_apply_generator": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "ryd_probabilities": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ryd_occupations": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ryd_observe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]),
    "ryd_ket_to_dm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ryd_outer_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ryd_outer_accumulate_dim": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p]),
    "ryd_accumulate": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "ryd_replay_samples": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_double, C.c_double, C.c_void_p, C.c_int32]),
    "ryd_get_stats": (C.c_int, [C.c_void_p, C.POINTER(RydStats)]),
    "ryd_reset_stats": (C.c_int, [C.c_void_p]),
    "ryd_set_kernel_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "ryd_get_kernel_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ryd_last_error": (C.c_char_p, []),
    "ryd_abi_version": (C.c_int, []),
}

_lib = None


def load() -> C.CDLL:
    """Load librydemu.so (built in-tree by ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    # dev probes (tools/) may load an alternative build of the same library (variant switches): only with the explicit
    # opt-in RYD_DEV=1 next to RYD_LIB - a stray environment variable must not swap the native library of a product run
    global LIB_PATH
    dev_lib = os.environ.get("RYD_LIB") if os.environ.get("RYD_DEV") == "1" else None
    LIB_PATH = dev_lib or LIB_PATH
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or "
            "make -C pulser_amd/csrc, same command). There is no CPU fallback."
        )
    try:
        # torch ships its own libamdhip64 (same SONAME); load it first so the
        # process has a single HIP runtime and torch device pointers are valid.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for symbol checks
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if not hasattr(lib, name) and dev_lib:
            continue  # dev probes against an older build of the library
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.ryd_abi_version() != RYD_ABI_VERSION:
        raise ImportError(
            f"librydemu ABI {lib.ryd_abi_version()} != binding {RYD_ABI_VERSION}"
        )
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        raise RydError(code, load().ryd_last_error().decode("utf-8", "replace"))
