"""Numeric restatement of ``pulser.NoiseModel`` (the fields the hot path reads).

Mirrors pulser-core/pulser/noise_model.py:175-520: same field names, defaults
and noise-type inference (``noise_types`` = the noise families whose
parameters are truthy, :401-405, minus doppler when ``disable_doppler``).  Any
object exposing these attributes (e.g. a real ``pulser.NoiseModel``) can be
passed wherever a NoiseModel is expected - only attribute access is used.
"""

from __future__ import annotations

import math
import warnings
from dataclasses import dataclass, field, fields
from typing import Any

import numpy as np

# pulser-core/pulser/constants.py:18-23 (env-overridable there; fixed here)
TRAP_WAVELENGTH = 0.85  # um
MASS = 1.45e-25  # kg
KB = 1.38e-23  # J/K
KEFF = 8.7  # um^-1

_NOISE_TYPE_PARAMS: dict[str, tuple[str, ...]] = {
    "leakage": ("with_leakage",),
    "doppler": ("temperature",),
    "register": ("trap_waist", "trap_depth"),
    "amplitude": ("laser_waist", "amp_sigma"),
    "detuning": ("detuning_sigma", "detuning_hf_psd", "detuning_hf_omegas"),
    "SPAM": ("p_false_pos", "p_false_neg", "state_prep_error"),
    "dephasing": ("dephasing_rate", "hyperfine_dephasing_rate"),
    "relaxation": ("relaxation_rate",),
    "depolarizing": ("depolarizing_rate",),
    "eff_noise": ("eff_noise_rates", "eff_noise_opers"),
    "dmm_sigma": ("dmm_sigma",),
    "dmm_crosstalk": ("detuning_map_spot_waist",),
}
_PARAM_TO_NOISE_TYPE = {
    p: nt for nt, params in _NOISE_TYPE_PARAMS.items() for p in params
}

# pulser-core/pulser/noise_model.py:101-114
LEGACY_DEFAULTS = {
    "runs": 15,
    "samples_per_run": 5,
    "state_prep_error": 0.005,
    "p_false_pos": 0.01,
    "p_false_neg": 0.05,
    "temperature": 50.0,
    "laser_waist": 175.0,
    "amp_sigma": 5e-2,
    "relaxation_rate": 0.01,
    "dephasing_rate": 0.05,
    "hyperfine_dephasing_rate": 1e-3,
    "depolarizing_rate": 0.05,
}


def doppler_sigma(temperature: float) -> float:
    """noise_model.py:127-133 (temperature in K)."""
    return KEFF * math.sqrt(KB * temperature / MASS)


def register_sigma_xy_z(
    temperature: float, trap_waist: float, trap_depth: float
) -> tuple[float, float]:
    """noise_model.py:136-171."""
    sxy = math.sqrt(temperature * trap_waist**2 / (4 * trap_depth))
    sz = math.pi / TRAP_WAVELENGTH * math.sqrt(2) * trap_waist * sxy
    return sxy, sz


def _to_tuple(obj: Any) -> Any:
    if isinstance(obj, np.ndarray):
        return tuple(_to_tuple(el) for el in obj)
    if isinstance(obj, (tuple, list)):
        return tuple(_to_tuple(el) for el in obj)
    return obj


_POSITIVE = {"dephasing_rate", "hyperfine_dephasing_rate", "relaxation_rate", "depolarizing_rate",
             "temperature", "detuning_sigma", "trap_waist"}
_STRICT_POSITIVE = {"runs", "samples_per_run", "laser_waist", "trap_depth", "detuning_map_spot_waist"}
_PROBABILITY_LIKE = {"state_prep_error", "p_false_pos", "p_false_neg", "amp_sigma", "dmm_sigma"}
_BOOLEAN = {"with_leakage", "disable_doppler"}


def _check_detuning_hf_noise(psd: tuple, freqs: tuple) -> None:
    """pulser/noise_model.py:539-583."""
    if (psd == ()) ^ (freqs == ()):
        raise ValueError("`detuning_hf_psd` and `detuning_hf_omegas` must either"
                         " both be empty tuples or both be provided.")
    if psd == ():
        return
    psd_a, freqs_a = np.asarray(psd), np.asarray(freqs)
    both = "`detuning_hf_psd` and `detuning_hf_omegas`"
    if psd_a.ndim != 1 or freqs_a.ndim != 1:
        raise ValueError(f"{both} are expected to be 1D tuples.")
    if psd_a.size != freqs_a.size:
        raise ValueError(f"{both} are expected to have the same length.")
    if psd_a.size <= 1:
        raise ValueError(f"{both} are expected to have length > 1.")
    if not (np.all(psd_a > 0) and np.all(freqs_a > 0)):
        raise ValueError(f"{both} are expected to have positive values.")
    if np.any(np.diff(freqs_a) < 0):
        raise ValueError("`detuning_hf_omegas` are expected to be monotonously growing.")


def check_eff_noise(rates: Any, opers: Any, check_contents: bool, with_leakage: bool) -> None:
    """pulser/noise_model.py:585-644: lengths and rate types always; when the noise
    type is active also non-empty, non-negative rates and 2-D operators of the
    single-qudit dimension (or one more, for the extra level of a larger basis)."""
    if len(opers) != len(rates):
        raise ValueError(
            f"The operators list length({len(opers)}) and rates list length"
            f"({len(rates)}) must be equal."
        )
    for rate in rates:
        if not isinstance(rate, (float, int)):
            raise TypeError(f"eff_noise_rates is a list of floats, it must not contain a {type(rate)}.")
    if not check_contents:
        return
    if not opers or not rates:
        raise ValueError("The effective noise parameters have not been filled.")
    if np.any(np.array(rates) < 0):
        raise ValueError("The provided rates must be greater than 0.")
    smallest = 3 if with_leakage else 2
    allowed = [(smallest, smallest), (smallest + 1, smallest + 1)]
    for op in opers:
        try:
            arr = np.array(op, dtype=complex)
        except (TypeError, ValueError) as e:
            raise TypeError(f"Operator {op!r} is not castable to a Numpy array.") from e
        if arr.ndim != 2:
            raise ValueError(f"Operator '{op!r}' is not a 2D array.")
        if arr.shape not in allowed:
            raise ValueError(
                f"With{'' if with_leakage else 'out'} leakage, operator's "
                f"shape must be {allowed[0]}, not {arr.shape}."
            )


@dataclass(frozen=True)
class NoiseModel:
    noise_types: tuple[str, ...] = field(init=False, default=())
    runs: int | None = None
    samples_per_run: int = 1
    state_prep_error: float = 0.0
    p_false_pos: float = 0.0
    p_false_neg: float = 0.0
    temperature: float = 0.0
    laser_waist: float | None = None
    amp_sigma: float = 0.0
    detuning_sigma: float = 0.0
    detuning_hf_psd: tuple = ()
    detuning_hf_omegas: tuple = ()
    relaxation_rate: float = 0.0
    dephasing_rate: float = 0.0
    trap_waist: float = 0.0
    trap_depth: float | None = None
    hyperfine_dephasing_rate: float = 0.0
    depolarizing_rate: float = 0.0
    eff_noise_rates: tuple = ()
    eff_noise_opers: tuple = ()
    with_leakage: bool = False
    disable_doppler: bool = False
    dmm_sigma: float = 0.0
    detuning_map_spot_waist: float | None = None

    def __post_init__(self) -> None:
        vals = {f.name: getattr(self, f.name) for f in fields(self) if f.init}
        for key in ("eff_noise_rates", "eff_noise_opers", "detuning_hf_psd", "detuning_hf_omegas"):
            vals[key] = _to_tuple(vals[key])
            object.__setattr__(self, key, vals[key])

        for p in _POSITIVE | _PROBABILITY_LIKE:  # noise_model.py:386-395
            try:
                vals[p] = float(vals[p])
            except (TypeError, ValueError):
                raise TypeError(f"{p} should be castable to float, not of type {type(vals[p])}.") from None
            object.__setattr__(self, p, vals[p])

        def truthy(v: Any) -> bool:
            if isinstance(v, tuple):
                return len(v) > 0
            return bool(v)

        types = {
            _PARAM_TO_NOISE_TYPE[p]
            for p, v in vals.items()
            if p in _PARAM_TO_NOISE_TYPE and truthy(v)
        }
        if "leakage" in types and "eff_noise" not in types:
            raise ValueError(
                "At least one effective noise operator must be defined to "
                "simulate leakage."
            )
        check_eff_noise(vals["eff_noise_rates"], vals["eff_noise_opers"], "eff_noise" in types,
                        bool(vals["with_leakage"]))
        _check_detuning_hf_noise(vals["detuning_hf_psd"], vals["detuning_hf_omegas"])
        relevant = self._find_relevant_params(types, vals["state_prep_error"], vals["amp_sigma"],
                                              vals["laser_waist"])
        if vals["runs"] is not None:
            warnings.warn(
                "Defining the number of emulation trajectories via 'NoiseModel.runs' is deprecated "
                "since pulser v1.7. Please favour using 'EmulationConfig.n_trajectories' instead.",
                category=DeprecationWarning, stacklevel=3)
        for p, v in vals.items():  # noise_model.py:646-675: only defined or relevant parameters
            if (v is None and p not in relevant) or (p == "runs" and v is None):
                continue
            if p in _POSITIVE and not v >= 0:
                raise ValueError(f"'{p}' must be greater than or equal to zero, not {v}.")
            if p in _STRICT_POSITIVE and not (v is not None and v > 0):
                raise ValueError(f"'{p}' must be greater than zero, not {v}.")
            if p in _PROBABILITY_LIKE and not 0 <= v <= 1:
                raise ValueError(f"'{p}' must be greater than or equal to zero and smaller than "
                                 f"or equal to one, not {v}.")
            if p in _BOOLEAN and not isinstance(v, bool):
                raise ValueError(f"'{p}' must be a boolean, not {v}.")
            if p == "samples_per_run" and v != 1:
                warnings.warn("Setting samples_per_run different to 1 is deprecated since pulser v1.6.",
                              DeprecationWarning, stacklevel=3)
        if "register" in types and (
            vals["trap_waist"] == 0.0 or vals["trap_depth"] is None or vals["temperature"] == 0.0
        ):
            raise ValueError(
                "trap_waist, trap_depth, and temperature must be defined in "
                "order to simulate register noise."
            )
        if vals["disable_doppler"]:
            types.discard("doppler")
        object.__setattr__(self, "noise_types", tuple(sorted(types)))
        defined = [p for p in relevant if truthy(vals[p])]
        for p, v in vals.items():  # noise_model.py:460-475
            unused = truthy(v) if p != "samples_per_run" else v != 1
            if p != "disable_doppler" and p not in relevant and unused:
                warnings.warn(
                    f"{p!r} is not used by any active noise type in {self.noise_types} when the "
                    f"only defined parameters are {defined}.", stacklevel=3)

    @staticmethod
    def _find_relevant_params(noise_types: Any, state_prep_error: float, amp_sigma: float,
                              laser_waist: float | None) -> set[str]:
        """noise_model.py:492-516: the parameters the active noise types read."""
        relevant: set[str] = set()
        for nt in noise_types:
            relevant.update(_NOISE_TYPE_PARAMS[nt])
            if nt == "register":
                relevant.add("temperature")
            if (nt in ("doppler", "detuning", "register", "dmm_sigma")
                    or (nt == "amplitude" and amp_sigma != 0.0)
                    or (nt == "SPAM" and state_prep_error != 0.0)):
                relevant.update(("runs", "samples_per_run"))
        if laser_waist is None:
            relevant.discard("laser_waist")
        return relevant

    # -- JSON abstract representation (pulser/noise_model.py:676-699) ----------
    _OPTIONAL_IN_ABSTR_REPR = ("detuning_sigma", "trap_waist", "trap_depth", "detuning_hf_psd",
                               "detuning_hf_omegas", "dmm_sigma", "detuning_map_spot_waist")

    def _to_abstract_repr(self) -> dict[str, Any]:
        defaults = {f.name: f.default for f in fields(self) if f.init}
        out: dict[str, Any] = {}
        for f in fields(self):
            value = getattr(self, f.name)
            if f.name in self._OPTIONAL_IN_ABSTR_REPR and defaults[f.name] == value:
                continue
            out[f.name] = value
        out.pop("disable_doppler")
        out.pop("with_leakage")
        rates, opers = out.pop("eff_noise_rates"), out.pop("eff_noise_opers")
        out["eff_noise"] = [[r, np.asarray(o).tolist()] for r, o in zip(rates, opers)]
        if "detuning_hf_psd" in out:
            psd, om = out.pop("detuning_hf_psd"), out.pop("detuning_hf_omegas")
            out["detuning_hf"] = [list(p) for p in zip(psd, om)]
        out["noise_types"] = list(out["noise_types"])
        return out

    @classmethod
    def _from_abstract_repr(cls, obj: dict[str, Any]) -> "NoiseModel":
        """pulser/json/abstract_repr/deserializer.py:440-505."""
        def cplx(v: Any) -> Any:
            if isinstance(v, list):
                return [cplx(e) for e in v]
            if isinstance(v, dict) and v.keys() == {"real", "imag"}:
                return v["real"] + 1j * v["imag"]
            return v

        obj = dict(obj)
        rates, opers = [], []
        for rate, oper in obj.pop("eff_noise"):
            rates.append(rate)
            opers.append(np.array(cplx(oper)))
        noise_types = list(obj.pop("noise_types"))
        disable_doppler = obj["temperature"] > 0 and "doppler" not in noise_types
        relevant: set[str] = set()
        for nt in noise_types + (["doppler"] if disable_doppler else []):
            relevant.update(_NOISE_TYPE_PARAMS[nt])
            if nt == "register":
                relevant.add("temperature")
            if (nt in ("doppler", "detuning", "register", "dmm_sigma")
                    or (nt == "amplitude" and obj["amp_sigma"] != 0.0)
                    or (nt == "SPAM" and obj["state_prep_error"] != 0.0)):
                relevant.update(("runs", "samples_per_run"))
        if obj.get("laser_waist") is None:
            relevant.discard("laser_waist")
        relevant -= {"eff_noise_rates", "eff_noise_opers", "with_leakage", "detuning_sigma",
                     "detuning_hf_psd", "detuning_hf_omegas", "dmm_sigma", "detuning_map_spot_waist"}
        psd, om = [], []
        for p_, f_ in obj.pop("detuning_hf", []):
            psd.append(p_)
            om.append(f_)
        nm = cls(**{p: obj[p] for p in relevant if p in obj},
                 eff_noise_rates=tuple(rates), eff_noise_opers=tuple(opers),
                 with_leakage="leakage" in noise_types, disable_doppler=disable_doppler,
                 detuning_hf_psd=tuple(psd), detuning_hf_omegas=tuple(om),
                 detuning_sigma=obj.get("detuning_sigma", 0), dmm_sigma=obj.get("dmm_sigma", 0),
                 detuning_map_spot_waist=obj.get("detuning_map_spot_waist"))
        if set(nm.noise_types) != set(noise_types):
            raise ValueError(f"Inconsistent noise model: {sorted(nm.noise_types)} != {sorted(noise_types)}")
        return nm

    def __repr__(self) -> str:
        shown = {f.name: getattr(self, f.name) for f in fields(self)
                 if f.init and getattr(self, f.name) not in (None, 0, 0.0, (), False)
                 or f.name == "samples_per_run"}
        args = ", ".join(f"{k}={v!r}" for k, v in shown.items())
        return f"NoiseModel(noise_types={self.noise_types!r}, {args})"


def has_shot_to_shot_except_spam(nm: Any) -> bool:
    """pulser-core/pulser/_hamiltonian_data/hamiltonian_data.py:89-104."""
    return (
        "doppler" in nm.noise_types
        or ("amplitude" in nm.noise_types and nm.amp_sigma != 0.0)
        or "detuning" in nm.noise_types
        or "register" in nm.noise_types
        or "dmm_sigma" in nm.noise_types
    )


def has_stochastic_noise(nm: Any) -> bool:
    """pulser-simulation/pulser_simulation/simulation.py:61-64."""
    return has_shot_to_shot_except_spam(nm) or (
        "SPAM" in nm.noise_types and nm.state_prep_error != 0
    )
