"""Backend-agnostic inputs of the hot path and their noise trajectories.

Restates, for Ising (ground-rydberg / digital, incl. DMM channels) and XY sequences, what
``pulser._hamiltonian_data.HamiltonianData`` does between the sampler and the
Hamiltonian (pulser-core/pulser/_hamiltonian_data/hamiltonian_data.py):

* ``SequenceInputs``   - the plain-array stand-in of ``SequenceSamples`` +
  register + device numbers;
* ``to_nested_dict``   - sampler/samples.py:524-621;
* ``HamiltonianData``  - noise-trajectory draws in the reference's exact order
  on the GLOBAL ``np.random`` stream (:782-911), SPAM-only dedup with
  ``Counter.most_common()`` (:795-835), noisy samples (:408-534), interaction
  matrix (:562-652), collapse-operator specs (:654-739).

Outputs are the problem bundles of ``pulser_amd.problem``.
"""

from __future__ import annotations

from collections import Counter
from dataclasses import dataclass, field, replace
from typing import Mapping, Any, Iterator, Sequence

import numpy as np

from .noise_model import (NoiseModel, doppler_sigma,
                          has_shot_to_shot_except_spam, register_sigma_xy_z)
from .problem import COORD_PRECISION

STATES_RANK = ["u", "d", "r", "g", "h", "x"]  # pulser/channels/base_channel.py
EIGENSTATES = {"ground-rydberg": ["r", "g"], "digital": ["g", "h"], "XY": ["u", "d"]}

SUPPORTED_NOISES = {
    "ising": {"amplitude", "detuning", "dephasing", "relaxation", "depolarizing",
              "doppler", "eff_noise", "SPAM", "leakage", "register", "dmm_sigma",
              "dmm_crosstalk"},
    "XY": {"dephasing", "depolarizing", "eff_noise", "SPAM", "leakage", "register"},
}


@dataclass
class Slot:
    """``_PulseTargetSlot`` (sampler/samples.py): samples [ti, tf) hit targets."""

    ti: int
    tf: int
    targets: tuple[int, ...]  # qubit indices


@dataclass
class ChannelInput:
    """One channel's samples (``ChannelSamples``, sampler/samples.py:95-246)."""

    name: str
    addressing: str  # "Global" | "Local"
    basis: str  # "ground-rydberg" | "digital" | "XY"
    amp: np.ndarray
    det: np.ndarray
    phase: np.ndarray
    slots: list[Slot] = field(default_factory=list)
    propagation_dir: tuple[float, float, float] | None = None
    # DMM channels (``DMMSamples``, sampler/samples.py:448-456): the detuning map
    # (sorted trap coordinates + weights, register/weight_maps.py:78-90), the
    # register positions it was sampled with (by qubit index; NaN = qubit unknown
    # to the map) and the spot waist of the samples.
    dmm_trap_coords: np.ndarray | None = None
    dmm_weights: np.ndarray | None = None
    dmm_qubit_coords: np.ndarray | None = None
    dmm_spot_waist: float | None = None
    # Sequence left in EOM mode (``eom_blocks[-1].tf is None``): extensions keep the
    # detuning at the block's detuning_off instead of zero (samples.py:170-180).
    final_detuning: float = 0.0

    @property
    def is_dmm(self) -> bool:
        return self.dmm_weights is not None

    def dmm_weight_map(self, spot_waist: float | None) -> np.ndarray:
        """``DetuningMap.get_qubit_weight_map`` (register/weight_maps.py:92-114):
        weight of the detuning on every qubit - the trap it sits on (within
        COORD_PRECISION), or with a spot waist the Gaussian-weighted sum over
        all traps (crosstalk)."""
        q = np.asarray(self.dmm_qubit_coords, float)
        traps = np.asarray(self.dmm_trap_coords, float)
        known = ~np.isnan(q).any(axis=1)
        dim = min(q.shape[1], traps.shape[1])
        diff = np.where(known[:, None, None], q[:, None, :dim] - traps[None, :, :dim], np.inf)
        dists = np.sqrt((diff * diff).sum(axis=2))
        if spot_waist:
            shape = np.exp(-(dists**2) / (2 * spot_waist**2))
        else:
            shape = dists < np.sqrt(2) * (10 ** (-6))
        w = shape @ np.asarray(self.dmm_weights, float)
        return np.where(known, w, 0.0)  # defaultdict(int) for qubits outside the map

    @property
    def duration(self) -> int:
        return len(self.amp)

    def extend_duration(self, new_duration: int) -> "ChannelInput":
        """``ChannelSamples.extend_duration`` (samples.py:152-200): pad amp with
        zeros, det with zeros (or detuning_off while in EOM mode), phase with its
        last value ('edge')."""
        extra = new_duration - self.duration
        if extra < 0:
            raise ValueError("Can't extend samples to a lower duration.")
        if extra == 0:
            return self
        pad = lambda a, mode: np.pad(np.asarray(a, float), (0, extra), mode=mode)  # noqa: E731
        phase = (
            pad(self.phase, "edge") if self.duration > 0 else np.zeros(new_duration)
        )
        det = np.pad(np.asarray(self.det, float), (0, extra), mode="constant",
                     constant_values=float(self.final_detuning))
        return replace(self, amp=pad(self.amp, "constant"), det=det, phase=phase)


@dataclass
class SequenceInputs:
    """Everything the path needs from sequence + register + device."""

    coords: np.ndarray  # float[N, 2 or 3], register order = tensor order
    qubit_ids: tuple[str, ...]
    channels: list[ChannelInput]
    interaction_coeff: float  # C6 (rad.um^6/us), devices/_device_datacls.py:382
    measurement: str | None = None
    slm_end: int = 0
    slm_targets: tuple[int, ...] = ()
    # XY mode (microwave channels): C3 coefficient and magnetic field
    interaction_coeff_xy: float | None = None  # devices/_device_datacls.py:392
    magnetic_field: tuple[float, float, float] | None = None

    def to_dict(self) -> dict[str, Any]:
        """Plain dict (arrays + scalars) for fixtures / IPC."""
        return {
            "coords": np.asarray(self.coords, float), "qubit_ids": tuple(self.qubit_ids),
            "interaction_coeff": float(self.interaction_coeff),
            "measurement": self.measurement or "", "slm_end": int(self.slm_end),
            "slm_targets": tuple(int(t) for t in self.slm_targets),
            "interaction_coeff_xy": -1.0 if self.interaction_coeff_xy is None else float(self.interaction_coeff_xy),
            "magnetic_field": np.zeros(0) if self.magnetic_field is None else np.asarray(self.magnetic_field, float),
            "channels": [
                {"name": c.name, "addressing": c.addressing, "basis": c.basis,
                 "amp": np.asarray(c.amp, float), "det": np.asarray(c.det, float),
                 "phase": np.asarray(c.phase, float),
                 "slots": [np.array([s.ti, s.tf] + list(s.targets), dtype=np.int64) for s in c.slots],
                 "propagation_dir": np.zeros(0) if c.propagation_dir is None else np.asarray(c.propagation_dir, float),
                 "dmm_trap_coords": np.zeros((0, 2)) if not c.is_dmm else np.asarray(c.dmm_trap_coords, float),
                 "dmm_weights": np.zeros(0) if not c.is_dmm else np.asarray(c.dmm_weights, float),
                 "dmm_qubit_coords": np.zeros((0, 2)) if not c.is_dmm else np.asarray(c.dmm_qubit_coords, float),
                 "dmm_spot_waist": -1.0 if c.dmm_spot_waist is None else float(c.dmm_spot_waist),
                 "is_dmm": bool(c.is_dmm), "final_detuning": float(c.final_detuning)}
                for c in self.channels
            ],
        }

    @classmethod
    def from_dict(cls, d: dict[str, Any]) -> "SequenceInputs":
        chans = []
        for c in d["channels"]:
            slots = [Slot(int(a[0]), int(a[1]), tuple(int(x) for x in a[2:])) for a in c["slots"]]
            pd = tuple(float(x) for x in c["propagation_dir"]) if len(c["propagation_dir"]) else None
            ch = ChannelInput(c["name"], c["addressing"], c["basis"], np.asarray(c["amp"], float),
                              np.asarray(c["det"], float), np.asarray(c["phase"], float), slots, pd)
            if c.get("is_dmm", False):
                sw = float(c["dmm_spot_waist"])
                ch = replace(ch, dmm_trap_coords=np.asarray(c["dmm_trap_coords"], float),
                             dmm_weights=np.asarray(c["dmm_weights"], float),
                             dmm_qubit_coords=np.asarray(c["dmm_qubit_coords"], float),
                             dmm_spot_waist=None if sw < 0 else sw)
            if float(c.get("final_detuning", 0.0)) != 0.0:
                ch = replace(ch, final_detuning=float(c["final_detuning"]))
            chans.append(ch)
        mf = tuple(float(x) for x in d["magnetic_field"]) if len(d["magnetic_field"]) else None
        xy = None if d["interaction_coeff_xy"] < 0 else float(d["interaction_coeff_xy"])
        return cls(np.asarray(d["coords"], float), tuple(d["qubit_ids"]), chans,
                   float(d["interaction_coeff"]), d["measurement"] or None, int(d["slm_end"]),
                   tuple(int(t) for t in d["slm_targets"]), xy, mf)

    @property
    def n_qudits(self) -> int:
        return len(self.qubit_ids)

    @property
    def max_duration(self) -> int:
        return max(ch.duration for ch in self.channels)

    @property
    def used_bases(self) -> set[str]:
        """``SequenceSamples.used_bases`` (samples.py:486-495): bases of the
        channels whose amplitude or detuning samples are not all zero."""
        return {ch.basis for ch in self.channels if not _is_empty(ch)}

    @property
    def in_xy(self) -> bool:
        return any(ch.basis == "XY" for ch in self.channels)

    def extend_duration(self, new_duration: int) -> "SequenceInputs":
        return replace(self, channels=[c.extend_duration(new_duration) for c in self.channels])

    def to_nested_dict(self, all_local: bool = False,
                       dmm: Mapping[str, tuple[float, float | None]] | None = None) -> dict[str, Any]:
        """sampler/samples.py:524-621.  In XY mode a global channel only reaches
        the SLM-masked atoms after ``slm_end``; before it the unmasked atoms get
        the samples as local entries (:574-587, :594-596).  A DMM channel is
        always distributed per qubit, its detuning times the qubit's map weight
        (:560-571, :598-608); ``dmm[name] = (factor, spot_waist)`` replaces the
        sampled detuning scale / spot waist the way
        ``_sample_with_trajectory`` does (hamiltonian_data.py:414-421)."""
        T = self.max_duration
        in_xy = self.in_xy
        d: dict[str, Any] = {"Global": {}, "Local": {}}

        def entry() -> dict[str, np.ndarray]:
            return {"amp": np.zeros(T), "det": np.zeros(T), "phase": np.zeros(T)}

        if in_xy:  # _prepare_dict(in_xy=True): the XY entries always exist
            d["Global"]["XY"] = entry()
            d["Local"]["XY"] = {}
        masked = set(self.slm_targets)
        for ch in self.channels:
            cs = ch.extend_duration(T)
            xy = ch.basis == "XY"
            weights = None
            det = cs.det
            if ch.is_dmm:
                factor, waist = (dmm or {}).get(ch.name, (1.0, ch.dmm_spot_waist))
                det = cs.det * factor
                weights = ch.dmm_weight_map(waist)
            if ch.addressing == "Global" and not all_local and not ch.is_dmm:
                g = d["Global"].setdefault(ch.basis, entry())
                start = self.slm_end if xy else 0
                g["amp"][start:] += cs.amp[start:]
                g["det"][start:] += cs.det[start:]
                g["phase"][start:] += cs.phase[start:]
                if start == 0:
                    continue
                loc = d["Local"].setdefault(ch.basis, {})
                for t in (set(cs.slots[0].targets) - masked if cs.slots else set()):
                    e = loc.setdefault(t, entry())
                    e["amp"][:start] += cs.amp[:start]
                    e["det"][:start] += cs.det[:start]
                    e["phase"][:start] += cs.phase[:start]
            else:
                loc = d["Local"].setdefault(ch.basis, {})
                for s in cs.slots:
                    for t in s.targets:
                        ti = s.ti
                        if xy and t in masked:
                            ti = max(ti, self.slm_end)
                        e = loc.setdefault(t, entry())
                        sl = slice(ti, s.tf)
                        e["amp"][sl] += cs.amp[sl]
                        e["det"][sl] += det[sl] if weights is None else det[sl] * weights[t]
                        e["phase"][sl] += cs.phase[sl]
        return d


def _is_empty(ch: ChannelInput) -> bool:
    """``ChannelSamples.is_empty`` (samples.py:202-214)."""
    return (np.count_nonzero(ch.amp) + np.count_nonzero(ch.det)) == 0


def single_global_channel(
    coords: np.ndarray,
    samples: dict[str, np.ndarray],
    interaction_coeff: float,
    basis: str = "ground-rydberg",
    name: str = "ising_global",
    prefix: str = "q",
    extended: bool = True,
) -> SequenceInputs:
    """Convenience: one global channel whose single pulse block spans the
    whole (un-extended) duration; ``samples`` already hold the extra trailing
    sample of simulation.py:173 when ``extended``."""
    n = len(coords)
    dur = len(samples["amp"]) - (1 if extended else 0)
    ch = ChannelInput(
        name, "Global", basis,
        np.asarray(samples["amp"], float), np.asarray(samples["det"], float),
        np.asarray(samples["phase"], float),
        slots=[Slot(0, dur, tuple(range(n)))],
    )
    return SequenceInputs(
        np.asarray(coords, float), tuple(f"{prefix}{i}" for i in range(n)), [ch],
        float(interaction_coeff),
    )


@dataclass
class NoiseTrajectory:
    """pulser/_hamiltonian_data/noise_trajectory.py:25-60."""

    bad_atoms: np.ndarray  # bool[N]
    doppler_detune: np.ndarray  # float[N]
    amp_fluctuations: dict[str, float]
    det_fluctuations: dict[str, float]
    det_phases: dict[str, np.ndarray]
    coords: np.ndarray
    interaction_matrix: np.ndarray
    reps: int = 1
    dmm_det_fluctuation: dict[str, float] = field(default_factory=dict)


def generate_detuning_fluctuations(
    nm: Any, det_cst_term: float, phases: np.ndarray, times: np.ndarray
) -> np.ndarray:
    """hamiltonian_data.py:132-169: delta_hf(t) + delta_sigma, times in ns."""
    det_hf = np.zeros_like(times, dtype=float)
    if nm.detuning_hf_psd:
        t = np.asarray(times) * 1e-3
        freqs = np.asarray(nm.detuning_hf_omegas)[1:]
        psd = np.asarray(nm.detuning_hf_psd)[1:]
        df = np.diff(nm.detuning_hf_omegas)
        amp = np.sqrt(2.0 * df * psd)
        arg = freqs[:, None] * t[None, :] + phases[:, None]
        det_hf = (amp[:, None] * np.cos(arg)).sum(axis=0)
    return det_cst_term + det_hf


def distances(coords: np.ndarray) -> np.ndarray:
    """hamiltonian_data.py:172-189 (cdist rounded to COORD_PRECISION)."""
    c = np.asarray(coords, float)
    diff = c[:, None, :] - c[None, :, :]
    return np.round(np.sqrt((diff**2).sum(-1)), COORD_PRECISION)


def finite_waist_amp_fraction(
    coords: Sequence[float], propagation_dir: Sequence[float], laser_waist: float
) -> float:
    """hamiltonian_data.py:758-780."""
    pos = np.zeros(3)
    pos[: len(coords)] = np.array(coords, float)
    u = np.array(propagation_dir, float)
    u = u / np.linalg.norm(u)
    dist = np.linalg.norm(pos - np.dot(pos, u) * u)
    return float(np.exp(-((dist / laser_waist) ** 2)))


class HamiltonianData:
    """Noise trajectories of one sequence (hamiltonian_data.py:192-943)."""

    def __init__(
        self,
        samples: SequenceInputs,
        noise_model: Any | None,
        n_trajectories: int | None,
    ) -> None:
        if samples.max_duration == 0:
            raise ValueError("SequenceSamples is empty.")
        self.samples = samples
        self.noise_model = noise_model if noise_model is not None else NoiseModel()
        self._check_noise_model()
        if n_trajectories is None:
            n_trajectories = 1
        nm = self.noise_model
        self.local_noises = True  # hamiltonian_data.py:259-273
        if set(nm.noise_types).issubset(
            {"dephasing", "relaxation", "SPAM", "depolarizing", "eff_noise", "leakage"}
        ):
            self.local_noises = "SPAM" in nm.noise_types and nm.state_prep_error > 0
        self.noise_trajectories = self._create_noise_trajectories(n_trajectories)

    # -- basis -------------------------------------------------------------
    @property
    def n_qudits(self) -> int:
        return self.samples.n_qudits

    @property
    def interaction_type(self) -> str:
        return "XY" if self.samples.in_xy else "ising"

    @property
    def basis_name(self) -> str:
        """hamiltonian_data.py:913-924."""
        used = self.samples.used_bases
        if len(used) == 0:
            name = "XY" if self.samples.in_xy else "ground-rydberg"
        elif len(used) == 1:
            name = list(used)[0]
        else:
            name = "all"
        if self.noise_model.with_leakage:
            name += "_with_error"
        return name

    @property
    def eigenbasis(self) -> list[str]:
        """hamiltonian_data.py:926-931 + SequenceSamples.eigenbasis."""
        used = self.samples.used_bases
        if len(used) == 0:
            used = {"XY" if self.samples.in_xy else "ground-rydberg"}
        states = set()
        for b in used:
            states.update(EIGENSTATES[b])
        if self.noise_model.with_leakage:
            states.add("x")
        return [s for s in STATES_RANK if s in states]

    def _check_noise_model(self) -> None:
        not_supported = set(self.noise_model.noise_types) - SUPPORTED_NOISES[self.interaction_type]
        if not_supported:
            raise NotImplementedError(
                f"Interaction mode '{self.interaction_type}' does not support "
                f"simulation of noise types: {', '.join(not_supported)}."
            )

    # -- collapse operators (hamiltonian_data.py:654-739) ---------------------
    def collapse_ops(self) -> tuple[list, dict]:
        nm = self.noise_model
        eigenbasis = self.eigenbasis
        names = ["I"] + [f"sigma_{a}{b}" for a in eigenbasis for b in eigenbasis]
        ops: list[tuple[Any, Any]] = []
        paulis: dict[str, list[tuple[complex, str]]] = {}
        if "dephasing" in nm.noise_types:
            rates = {"d": nm.dephasing_rate, "r": nm.dephasing_rate, "h": nm.hyperfine_dephasing_rate}
            for state in eigenbasis:
                if state in rates:
                    ops.append((np.sqrt(2 * rates[state]), f"sigma_{state}{state}"))
        if "relaxation" in nm.noise_types:
            if "sigma_gr" not in names:
                raise ValueError(
                    "'relaxation' noise requires addressing of the 'ground-rydberg' basis."
                )
            ops.append((np.sqrt(nm.relaxation_rate), "sigma_gr"))
        if "depolarizing" in nm.noise_types:
            if "all" in self.basis_name:
                raise NotImplementedError("Cannot include depolarizing noise in all-basis.")
            b, a = eigenbasis[:2]
            paulis["x"] = [(1, f"sigma_{a}{b}"), (1, f"sigma_{b}{a}")]
            paulis["y"] = [(1j, f"sigma_{a}{b}"), (-1j, f"sigma_{b}{a}")]
            paulis["z"] = [(1, f"sigma_{b}{b}"), (-1, f"sigma_{a}{a}")]
            coeff = np.sqrt(nm.depolarizing_rate / 4)
            for label in paulis:
                ops.append((coeff, label))
        if "eff_noise" in nm.noise_types:
            d = len(eigenbasis)
            for id_, rate in enumerate(nm.eff_noise_rates):
                op = np.array(nm.eff_noise_opers[id_])
                if op.shape != (d, d):
                    raise ValueError(
                        f"Incompatible shape for effective noise operator n°{id_}. "
                        f"Operator {op} should be of shape {(d, d)}."
                    )
                ops.append((np.sqrt(rate), op))
        return ops, paulis

    # -- interaction (hamiltonian_data.py:562-652) ------------------------------
    def interaction_matrix(self, coords: np.ndarray, bad_atoms: np.ndarray) -> np.ndarray:
        # without register noise every trajectory has the same positions: build the pair loop once
        key = (np.asarray(coords, float).tobytes(), np.asarray(bad_atoms, bool).tobytes())
        cache = self.__dict__.setdefault("_inter_cache", {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = self._interaction_matrix(coords, bad_atoms)
        return cache[key].copy()

    def _interaction_matrix(self, coords: np.ndarray, bad_atoms: np.ndarray) -> np.ndarray:
        n = self.n_qudits
        d = distances(coords)
        is_xy = self.interaction_type == "XY"
        inter = np.zeros((2 if is_xy else 1, n, n))
        if is_xy:  # C3 (1 - 3 cos^2 theta) / r^3 with theta to the magnetic field (:589-611)
            if self.samples.magnetic_field is None or self.samples.interaction_coeff_xy is None:
                raise ValueError("XY mode needs 'magnetic_field' and 'interaction_coeff_xy'.")
            mag = np.asarray(self.samples.magnetic_field, float)
            mag_norm = np.linalg.norm(mag)
            assert mag_norm > 0, "There must be a magnetic field in XY mode."
            pos = np.asarray(coords, float)
            if pos.shape[1] == 2:
                pos = np.column_stack((pos, np.zeros(n)))
            for i in range(n):
                for j in range(i + 1, n):
                    diff = pos[i] - pos[j]
                    cosine = np.dot(diff, mag) / (np.linalg.norm(diff) * mag_norm)
                    inter[0, i, j] = inter[0, j, i] = (
                        self.samples.interaction_coeff_xy * (1 - 3 * cosine**2) / d[i, j] ** 3
                    )
        for i in range(n):
            for j in range(i + 1, n):
                inter[-1, i, j] = inter[-1, j, i] = self.samples.interaction_coeff / d[i, j] ** 6
        bad = np.asarray(bad_atoms, bool)
        inter[:, bad.reshape(1, -1) | bad.reshape(-1, 1)] = 0.0
        return inter

    # -- trajectories (hamiltonian_data.py:782-911) -----------------------------
    def _create_noise_trajectories(self, ntrajs: int) -> list[NoiseTrajectory]:
        nm = self.noise_model
        n = self.n_qudits
        chans = [c.name for c in self.samples.channels]
        out: list[NoiseTrajectory] = []
        if not has_shot_to_shot_except_spam(nm):
            configs = Counter(
                "".join(
                    (np.random.uniform(size=n) < nm.state_prep_error).astype(int).astype(str)
                )
                for _ in range(ntrajs)
            ).most_common()
            for bool_string, reps in configs:
                bad = np.array([c == "1" for c in bool_string])
                out.append(
                    NoiseTrajectory(
                        bad, np.zeros(n),
                        {c: 1.0 for c in chans}, {c: 0.0 for c in chans},
                        {c: np.array(0.0) for c in chans},
                        self.samples.coords, self.interaction_matrix(self.samples.coords, bad),
                        reps, {c: 1.0 for c in chans},
                    )
                )
            return out
        # The reference creates this dict ONCE, outside the trajectory loop
        # (hamiltonian_data.py:794), and hands the same object to every
        # NoiseTrajectory: the draws advance the RNG per trajectory, but every
        # trajectory ends up with the factors drawn last.  Kept as is.
        dmm_f: dict[str, float] = {}
        is_dmm = {c.name: c.is_dmm for c in self.samples.channels}
        for _ in range(ntrajs):
            amp_f: dict[str, float] = {}
            det_f: dict[str, float] = {}
            det_ph: dict[str, np.ndarray] = {}
            coords = self.samples.coords
            if "SPAM" in nm.noise_types and nm.state_prep_error > 0:
                bad = np.random.uniform(size=n) < nm.state_prep_error
            else:
                bad = np.zeros(n, bool)
            if "doppler" in nm.noise_types:
                dop = np.random.normal(0, doppler_sigma(nm.temperature * 1e-6), size=n)
            else:
                dop = np.zeros(n)
            for c in chans:
                amp_f[c] = max(0, np.random.normal(1.0, nm.amp_sigma))
                det_f[c] = np.random.normal(0.0, nm.detuning_sigma) if nm.detuning_sigma else 0.0
                if nm.detuning_hf_omegas:
                    det_ph[c] = np.random.uniform(0.0, 2 * np.pi, size=len(nm.detuning_hf_omegas) - 1)
                else:
                    det_ph[c] = np.array(0.0)
                if nm.dmm_sigma and is_dmm[c]:  # :880-888
                    dmm_f[c] = max(0, np.random.normal(1.0, nm.dmm_sigma))
                else:
                    dmm_f[c] = 1.0
            if "register" in nm.noise_types:
                sxy, sz = register_sigma_xy_z(nm.temperature, nm.trap_waist, float(nm.trap_depth))
                pos = np.asarray(coords, float)
                if pos.shape[1] == 2:
                    pos = np.column_stack((pos, np.zeros(n)))
                narr_xy = np.random.normal(0, sxy, (n, 2))
                narr_z = np.random.normal(0, sz, n)
                coords = pos + np.column_stack((narr_xy, narr_z))
            out.append(
                NoiseTrajectory(bad, dop, amp_f, det_f, det_ph, coords,
                                self.interaction_matrix(coords, bad), 1, dmm_f)
            )
        return out

    # -- noisy samples (hamiltonian_data.py:408-534) ----------------------------
    def nested_samples(self, traj: NoiseTrajectory) -> dict[str, Any]:
        nm = self.noise_model
        if not self.local_noises:  # the noiseless samples as they are (:532-533)
            return self.samples.to_nested_dict(all_local=False)
        dmm = {c.name: (traj.dmm_det_fluctuation.get(c.name, 1.0), nm.detuning_map_spot_waist)
               for c in self.samples.channels if c.is_dmm}  # :414-421
        d = self.samples.to_nested_dict(all_local=True, dmm=dmm)
        T = self.samples.max_duration
        for ch in self.samples.channels:
            loc = d["Local"][ch.basis]
            det_fl = generate_detuning_fluctuations(
                nm, traj.det_fluctuations[ch.name], traj.det_phases[ch.name], np.arange(0, T, 1)
            )
            for slot in ch.slots:
                for q in slot.targets:
                    sl = slice(slot.ti, slot.tf)
                    if "doppler" in nm.noise_types:
                        loc[q]["det"][sl] += traj.doppler_detune[q]
                    if "amplitude" in nm.noise_types:
                        frac = traj.amp_fluctuations[ch.name]
                        if nm.laser_waist is not None and ch.addressing == "Global":
                            prop = ch.propagation_dir or (0.0, 1.0, 0.0)
                            frac *= finite_waist_amp_fraction(
                                tuple(traj.coords[q]), tuple(prop), nm.laser_waist
                            )
                        loc[q]["amp"][sl] *= frac
                    if "detuning" in nm.noise_types:
                        loc[q]["det"][sl] += det_fl[sl]
        for basis in d["Local"]:  # badly prepared atoms: everything zeroed (:507-509)
            for q, vals in d["Local"][basis].items():
                if traj.bad_atoms[q]:
                    for qty in ("amp", "det", "phase"):
                        vals[qty] *= 0.0
        return d

    def problem(self, traj: NoiseTrajectory, sampling_rate: float) -> dict[str, Any]:
        ops, paulis = self.collapse_ops()
        return {
            "n_qudits": self.n_qudits,
            "qubit_ids": tuple(self.samples.qubit_ids),
            "coords": np.asarray(traj.coords, float),
            "eigenbasis": self.eigenbasis,
            "basis_name": self.basis_name,
            "interaction_type": self.interaction_type,
            "duration": int(self.samples.max_duration),
            "sampling_rate": float(sampling_rate),
            "samples": self.nested_samples(traj),
            "interaction_matrix": traj.interaction_matrix,
            "bad_atoms": np.asarray(traj.bad_atoms, bool),
            "collapse_ops": ops,
            "depolarizing_pauli_2ds": paulis,
            "slm_end": int(self.samples.slm_end),
            "slm_targets": tuple(self.samples.slm_targets),
            "reps": int(traj.reps),
        }

    def problems(self, sampling_rate: float) -> Iterator[dict[str, Any]]:
        for traj in self.noise_trajectories:
            yield self.problem(traj, sampling_rate)


    # -- factored lowering: trajectories -> device tables without arrays ------
    def factorable(self) -> bool:
        """True when every trajectory's noisy samples are the shared channel
        samples scaled / offset per (trajectory, atom): each atom driven by one
        channel only, 2-level basis (the high-frequency detuning noise factors
        too: cos / sin series shared by the batch, per-trajectory amplitudes)."""
        if len(self.eigenbasis) != 2 or self.interaction_type != "ising":
            return False
        # a detuning-map modulator weights its detuning per qubit (DMM weights,
        # dmm_sigma factor, crosstalk spot waist: hamiltonian_data.py:414-421,
        # 880-888; samples.py DMMSamples) - that is not a (series, scale) pair of
        # the shared channel samples, so those sequences take lower(problem(...))
        if any(ch.is_dmm for ch in self.samples.channels):
            return False
        seen: set[tuple[str, int]] = set()
        for ch in self.samples.channels:
            targets = {t for s in ch.slots for t in s.targets}
            for t in targets:
                if (ch.basis, t) in seen:
                    return False
                seen.add((ch.basis, t))
        return True

    def device_tables(self, trajs: Sequence[NoiseTrajectory], sampling_rate: float) -> Any:
        """``pulser_amd.terms.DeviceTables`` for a batch of trajectories.

        Mathematically identical to lowering ``self.problem(traj)`` for every
        trajectory (splines are linear in their knots), but the per-trajectory
        noise of hamiltonian_data.py:408-534 is kept in factored form: the
        spline tables hold the shared (channel, target-mask) series once and each
        (trajectory, atom) descriptor carries the amplitude factor, the doppler /
        detuning offset (times the slot mask) and the bad-atom zeroing.
        """
        from .terms import (DESC_DTYPE, DTERM_DTYPE, DeviceTables, _SeriesPool,
                            adapt_to_sampling_rate, local_collapse_ops, local_dissipator, lower,
                            sampling_times)
        from scipy.interpolate import CubicSpline

        if not self.factorable():
            return lower([self.problem(t, sampling_rate) for t in trajs])
        nm = self.noise_model
        n, T = self.n_qudits, self.samples.max_duration
        basis_name = self.basis_name
        tknots = sampling_times(T, sampling_rate)
        pool = _SeriesPool()
        # high-frequency detuning noise (:132-169): sum_f A_f cos(w_f t + phi_f) inside the
        # slots = A_f cos(phi_f) [m cos(w_f t)] - A_f sin(phi_f) [m sin(w_f t)]
        hf = bool(nm.detuning_hf_psd) and "detuning" in nm.noise_types
        if hf:
            om = np.asarray(nm.detuning_hf_omegas, float)
            hf_freqs = om[1:]
            hf_amp = np.sqrt(2.0 * np.diff(om) * np.asarray(nm.detuning_hf_psd, float)[1:])
            t_us = np.arange(0, T, 1) * 1e-3
        hf_series: dict[int, list[tuple[int, int]]] = {}
        # per atom: (channel, drive series, det series, mask series)
        per_atom: dict[int, tuple[Any, int, int, int]] = {}
        for ch in self.samples.channels:
            cs = ch.extend_duration(T)
            masks: dict[int, np.ndarray] = {}
            for s in cs.slots:
                for t in s.targets:
                    masks.setdefault(t, np.zeros(T))[s.ti:s.tf] = 1.0
            for t, m in masks.items():
                c = 0.5 * (m * cs.amp) * np.exp(-1j * (m * cs.phase))
                di = pool.add(adapt_to_sampling_rate(c, sampling_rate, T))
                ti = pool.add(adapt_to_sampling_rate(m * cs.det, sampling_rate, T))
                mi = pool.add(adapt_to_sampling_rate(m, sampling_rate, T))
                per_atom[t] = (ch, di, ti, mi)
                if hf:
                    hf_series[t] = [
                        (pool.add(adapt_to_sampling_rate(m * np.cos(w * t_us), sampling_rate, T)),
                         pool.add(adapt_to_sampling_rate(m * np.sin(w * t_us), sampling_rate, T)))
                        for w in hf_freqs
                    ]
        desc = np.zeros((len(trajs), n), dtype=DESC_DTYPE)
        desc["drive_series"] = desc["det_series"] = desc["off_series"] = -1
        dterms: list[np.ndarray] = []
        n_dterms = 0
        hf_index: dict[tuple, int] = {}
        # per atom: the (cos, sin) series ids interleaved, and which of them exist
        hf_ids = {k: (np.array(v, dtype=np.int32).reshape(-1),
                      np.array(v, dtype=np.int32).reshape(-1) >= 0) for k, v in hf_series.items()}
        mats = []
        local = self.local_noises
        n_b = len(trajs)
        # the descriptors column by column (round 6: 512 trajectories x 12 atoms of field-by-field writes into the
        # structured array were 20 - 40 ms of Python on the critical path of an ensemble's first block)
        bad_all = np.array([np.asarray(tr.bad_atoms, bool) for tr in trajs]).reshape(n_b, n)
        waist = local and "amplitude" in nm.noise_types and nm.laser_waist is not None
        for k, (ch, di, ti, mi) in per_atom.items():
            frac = np.ones(n_b)
            off = np.zeros(n_b)
            live = np.ones(n_b, bool)
            if local:
                live = ~bad_all[:, k]  # amp, det and phase of a badly prepared atom are zeroed (:507-509)
                if "amplitude" in nm.noise_types:
                    frac = np.array([tr.amp_fluctuations[ch.name] for tr in trajs], dtype=float)
                    if waist and ch.addressing == "Global":
                        prop = tuple(ch.propagation_dir or (0.0, 1.0, 0.0))
                        frac = frac * np.array([finite_waist_amp_fraction(tuple(tr.coords[k]), prop, nm.laser_waist)
                                                for tr in trajs])
                if "doppler" in nm.noise_types:
                    off = off + np.array([tr.doppler_detune[k] for tr in trajs], dtype=float)
                if "detuning" in nm.noise_types:
                    off = off + np.array([float(tr.det_fluctuations[ch.name]) for tr in trajs])
            col = desc[:, k]
            if di >= 0:
                m = live & (frac != 0.0)
                col["drive_series"][m] = di
                col["drive_scale"][m] = frac[m]
            if ti >= 0:
                col["det_series"][live] = ti
                col["det_scale"][live] = 1.0
            if mi >= 0:
                m = live & (off != 0.0)
                col["off_series"][m] = mi
                col["off_scale"][m] = off[m]
        for b, tr in enumerate(trajs if (hf and local) else ()):
            if True:
                for k, (ch, di, ti, mi) in per_atom.items():
                    if tr.bad_atoms[k]:
                        continue
                    d = desc[b, k]
                    phases = np.atleast_1d(np.asarray(tr.det_phases[ch.name], float))
                    ids, keep = hf_ids[k]
                    key = (b, ch.name, ids.tobytes())
                    if key in hf_index:  # atoms sharing channel and slot mask share the list
                        d["extra"] = hf_index[key]
                    elif keep.any():
                        sc = np.empty(2 * len(hf_amp))
                        sc[0::2] = hf_amp * np.cos(phases)
                        sc[1::2] = -hf_amp * np.sin(phases)
                        block = np.zeros(int(keep.sum()), dtype=DTERM_DTYPE)
                        block["series"], block["scale"] = ids[keep], sc[keep]
                        block["remaining"] = np.arange(len(block) - 1, -1, -1)
                        dterms.append(block)
                        d["extra"] = hf_index[key] = n_dterms + 1
                        n_dterms += len(block)
        # interaction matrices of the batch at once: diagonal, badly prepared atoms' rows and columns zeroed
        u_all = np.array([np.asarray(tr.interaction_matrix, dtype=float)[-1] for tr in trajs]).reshape(n_b, n, n)
        u_all[:, np.arange(n), np.arange(n)] = 0.0
        keep_pair = ~bad_all[:, :, None] & ~bad_all[:, None, :]
        u_all *= keep_pair
        if "digital" in basis_name:
            u_all[:] = 0.0
        else:
            u_all[(n - bad_all.sum(axis=1)) <= 1] = 0.0
        mats = list(u_all)
        if not pool.arrays:
            pool.arrays.append(np.zeros(len(tknots), dtype=np.complex128))
        pp = np.empty((len(pool.arrays), len(tknots) - 1, 4), dtype=np.complex128)
        for i, knots in enumerate(pool.arrays):
            pp[i] = np.transpose(CubicSpline(tknots, knots, bc_type="not-a-knot").c, (1, 0))
        shared = all(np.array_equal(mats[0], m) for m in mats[1:])
        ops, paulis = self.collapse_ops()
        return DeviceTables(
            n_qubits=n, batch=len(trajs),
            tknots=np.ascontiguousarray(tknots, dtype=np.float64), pp=np.ascontiguousarray(pp),
            desc=desc, interaction=np.ascontiguousarray(np.stack(mats[:1] if shared else mats)),
            dissipator=local_dissipator(ops, self.eigenbasis, paulis),
            series_knots=pool.arrays,
            collapse_local=local_collapse_ops(ops, self.eigenbasis, paulis),
            dterms=np.concatenate(dterms) if dterms else None,
        )
