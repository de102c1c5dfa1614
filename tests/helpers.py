"""Shared helpers for the parity tests (oracle = checker only)."""
from __future__ import annotations

import os

import numpy as np

from pulser_amd import problem as P

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# Sequence budget of the split-operator step-size controller (host_split.hpp: kSplitTolTotal): since round 6 a bound on the
# 2-NORM of the accumulated error (sum of the local 2-norms), 0.8 of the parity bar of 1e-7 on every amplitude.
SPLIT_BUDGET = 8e-8

def load_fixture(name):
    prob, extra = P.load_problem(os.path.join(GOLDEN, name))
    return prob, extra


def with_anneal_samples(prob):
    """cfg2/cfg3 fixtures store no samples: regenerate them synthetically."""
    s = P.anneal_samples()
    prob = dict(prob)
    prob["samples"] = {"Global": {"ground-rydberg": s}, "Local": {}}
    return prob


def blockade_radius():
    return (P.C6_LEVEL70 / (4 * 2 * np.pi / 2)) ** (1 / 6)


def chain_problem(n, collapse_ops=None):
    coords = P.register_coords(P.square_rect(1, n), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=collapse_ops)


def local_problem(n, seed=0, duration=401, collapse_ops=None, paulis=None):
    """Random smooth per-qubit drives/detunings/phases (Local addressing)."""
    rng = np.random.default_rng(seed)
    t = np.arange(duration) / 1000.0
    coords = P.register_coords(P.square_rect(1, n), 7.0) + rng.normal(0, 0.3, (n, 2))
    prob = P.make_ising_problem(coords, {"amp": np.zeros(duration), "det": np.zeros(duration), "phase": np.zeros(duration)})
    loc = {}
    for q in range(n):
        a, b, c = rng.uniform(2, 12, 3)
        loc[q] = {
            "amp": a * (1 + 0.5 * np.sin(2 * np.pi * (q + 1) * t / t[-1])),
            "det": b * np.cos(3 * t + q) - c,
            "phase": 0.3 * q + 0.8 * np.sin(5 * t),
        }
    prob["samples"] = {"Global": {}, "Local": {"ground-rydberg": loc}}
    prob["collapse_ops"] = list(collapse_ops or [])
    prob["depolarizing_pauli_2ds"] = dict(paulis or {})
    return prob


DEPOL_PAULIS = {
    "x": [(1, "sigma_gr"), (1, "sigma_rg")],
    "y": [(1j, "sigma_gr"), (-1j, "sigma_rg")],
    "z": [(1, "sigma_rr"), (-1, "sigma_gg")],
}


def rand_state(dim, seed):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=dim) + 1j * rng.normal(size=dim)
    return v / np.linalg.norm(v)


def tight_density_matrices(extra, n):
    """cfg3 fixtures written by ``gen_cfg3_tight`` keep the upper triangle only."""
    D = 2**n
    iu = np.triu_indices(D)
    out = []
    for vals in np.asarray(extra["oracle_states_tight_triu"]):
        rho = np.zeros((D, D), complex)
        rho[iu] = vals
        low = rho.conj().T.copy()
        low[np.diag_indices(D)] = 0.0
        out.append(rho + low)
    return np.stack(out)


def sketch_rows(D, rows=32, seed=11):
    """The rows / probe vectors of ``make_fixtures.sketch_density_matrices`` (same seeded draws)."""
    rng = np.random.default_rng(seed)
    return np.unique(np.concatenate([[0, D - 1], rng.choice(D, min(D, rows) - 2, replace=False)]))


def sketch_probes(D, probes=4, seed=11):
    rng = np.random.default_rng(seed + 1)
    return rng.standard_normal((D, probes)) + 1j * rng.standard_normal((D, probes))


def sketch_errors(rho, extra, k):
    """Max-abs deviation of ``rho`` from the tight oracle at stored time ``k`` over everything the fixture
    keeps: 32 full rows, the diagonal, rho @ (4 Gaussian probes) - which sees an error anywhere - and purity."""
    D = rho.shape[0]
    sk = extra["sketch"]
    rows = sketch_rows(D, sk["rows"], sk["seed"])
    if sk.get("keep"):  # (12 atoms: every 4th of the 32 drawn rows - a 2^12-entry row is 64 KiB per stored time)
        rows = rows[np.round(np.linspace(0, len(rows) - 1, int(sk["keep"]))).astype(int)]
    assert np.array_equal(rows, np.asarray(extra["oracle_rows"]))
    V = sketch_probes(D, sk["probes"], sk["seed"])
    return {
        "rows": float(np.max(np.abs(rho[rows] - np.asarray(extra["oracle_rows_tight"])[k]))),
        "diag": float(np.max(np.abs(np.diag(rho) - np.asarray(extra["oracle_diag_tight"])[k]))),
        "probes": float(np.max(np.abs(rho @ V - np.asarray(extra["oracle_probe_products_tight"])[k]))),
        "purity": float(abs(np.vdot(rho, rho).real - np.asarray(extra["oracle_purity_tight"])[k])),
    }


# ---------------------------------------------------------------------------------------------------------------------
# Seeded random pulse sequences from the waveform families Pulser ships (pulser-core/pulser/waveforms.py: Constant,
# Ramp, Blackman, Kaiser, Interpolated (PCHIP), Composite; EOM-style square pulses with 1-ns edges and a detuning that
# jumps with them; back-to-back pulses with phase jumps; delays) - restated in NumPy / SciPy, no pulser import: the fuzz
# of the step-size controller (tests/test_gpu_fuzz.py, tools/fuzz_ctrl.py) runs on the GPU box.
# ---------------------------------------------------------------------------------------------------------------------
def _wf_constant(n, v):
    return np.full(n, float(v))


def _wf_ramp(n, a, b):
    return P.ramp_samples(n, a, b) if n > 1 else np.full(n, float(a))


def _wf_blackman(n, peak):
    w = np.clip(np.blackman(n), 0, np.inf)
    return peak * w / max(w.max(), 1e-300)


def _wf_kaiser(n, peak, beta=14.0):
    w = np.kaiser(n, beta)  # KaiserWaveform: np.kaiser window (waveforms.py), default beta 14
    return peak * w / max(w.max(), 1e-300)


def _wf_interpolated(rng, n, lo, hi, first=None, last=None):
    """InterpolatedWaveform's default interpolator (PchipInterpolator through 3 - 6 equally spaced values)."""
    from scipy.interpolate import PchipInterpolator

    k = int(rng.integers(3, 7))
    vals = rng.uniform(lo, hi, k)
    if first is not None:
        vals[0] = first
    if last is not None:
        vals[-1] = last
    return PchipInterpolator(np.linspace(0, 1, k), vals)(np.linspace(0, 1, n))


def random_pulse_samples(rng, duration, omega_max=None, det_max=None, complex_phase=True):
    """{"amp", "det", "phase"} of ``duration + 1`` samples (the extra trailing sample of simulation.py:173: amp = det = 0)
    of a random sequence of pulses on one global Rydberg channel."""
    omega_max = float(rng.uniform(3.0, 30.0)) if omega_max is None else omega_max
    det_max = float(rng.uniform(5.0, 80.0)) if det_max is None else det_max
    amp, det, ph = [], [], []
    t = 0
    phase = 0.0
    while t < duration:
        n = int(min(duration - t, rng.integers(16, max(17, min(1500, duration)))))
        if duration - t - n < 16:
            n = duration - t
        kind = rng.choice(["constant", "ramp", "blackman", "kaiser", "interp", "delay", "eom", "composite"])
        peak = float(rng.uniform(0.2, 1.0)) * omega_max
        if kind == "delay":
            a = np.zeros(n)
        elif kind == "constant" or kind == "eom":
            a = _wf_constant(n, peak)
        elif kind == "ramp":
            lo, hi = sorted(rng.uniform(0.0, 1.0, 2) * omega_max)
            a = _wf_ramp(n, lo, hi) if rng.random() < 0.5 else _wf_ramp(n, hi, lo)
        elif kind == "blackman":
            a = _wf_blackman(n, peak)
        elif kind == "kaiser":
            a = _wf_kaiser(n, peak, float(rng.uniform(4.0, 16.0)))
        elif kind == "interp":
            a = np.clip(_wf_interpolated(rng, n, 0.0, omega_max, first=0.0 if rng.random() < 0.5 else None,
                                         last=0.0 if rng.random() < 0.5 else None), 0.0, None)
        else:  # composite: rise - plateau - fall
            k1 = max(1, n // 4)
            a = np.concatenate([_wf_ramp(k1, 0.0, peak), _wf_constant(n - 2 * k1, peak), _wf_ramp(k1, peak, 0.0)])
        dk = rng.choice(["constant", "ramp", "interp"])
        if kind == "eom":  # square pulse: the detuning switches with the amplitude (EOM mode: detuning_on)
            d = _wf_constant(n, rng.uniform(-1.0, 1.0) * det_max)
        elif dk == "constant":
            d = _wf_constant(n, rng.uniform(-1.0, 1.0) * det_max)
        elif dk == "ramp":
            d = _wf_ramp(n, *(rng.uniform(-1.0, 1.0, 2) * det_max))
        else:
            d = _wf_interpolated(rng, n, -det_max, det_max)
        if complex_phase and rng.random() < 0.4:  # a phase jump between back-to-back pulses
            phase = float(rng.uniform(0.0, 2 * np.pi))
        amp.append(a)
        det.append(d)
        ph.append(np.full(n, phase))
        t += n
    amp = np.concatenate(amp + [[0.0]])
    det = np.concatenate(det + [[0.0]])
    ph = np.concatenate(ph + [[ph[-1][-1]]])
    assert len(amp) == duration + 1
    return {"amp": amp, "det": det, "phase": ph}


def random_register(rng, n):
    """coords of an n-atom chain / two-row triangular / rectangular register, spacing 4.5 - 10 um."""
    spacing = float(rng.uniform(4.5, 10.0))
    kinds = ["chain"]
    if n % 2 == 0:
        kinds += ["tri", "rect"]
    kind = rng.choice(kinds)
    if kind == "chain":
        lay = P.square_rect(1, n)
    elif kind == "tri":
        lay = P.triangular_rect(2, n // 2)
    else:
        r = 4 if n == 16 else 2
        lay = P.square_rect(r, n // r)
    return P.register_coords(lay, spacing), spacing, str(kind)


def fuzz_case(seed, n_atoms=None):
    """One case of the controller fuzz: (problems of one batch, description).  `n_atoms`: the same draws on a smaller
    register (8 - 11 atoms: the sizes a tight CPU oracle integrates in seconds, tests/golden/make_fuzz_fixtures.py)."""
    rng = np.random.default_rng(10_000 + seed)
    n = int(rng.choice([12, 12, 13, 13, 14, 14, 16]))
    if n_atoms is not None:
        n = int(n_atoms)
    dur_cap = {12: 4000, 13: 2500, 14: 1500, 16: 300}.get(n, 4000)
    duration = int(np.exp(rng.uniform(np.log(100), np.log(dur_cap))))
    batch = 1 if n == 16 else int(rng.choice([1, 1, 2, 4]))
    coords, spacing, kind = random_register(rng, n)
    cplx = bool(rng.random() < 0.5)
    probs = [P.make_ising_problem(coords, random_pulse_samples(rng, duration, complex_phase=cplx)) for _ in range(batch)]
    return probs, f"seed {seed}: {n} atoms ({kind}, {spacing:.2f} um), {duration} ns, batch {batch}, phases {cplx}"


# -- general-path problems shared by tests/test_gpu_general_free.py and tools/general_bench.py ---------------
def three_level_problem(n, T=201, seed=5):
    rng = np.random.default_rng(seed)
    lay = P.square_rect(3, 3) if n == 9 else P.square_rect(2, n // 2)
    coords = P.register_coords(lay, 6.5)
    t = np.arange(T) / 1000.0
    prob = P.make_ising_problem(coords, {"amp": 6.0 + 2.0 * np.sin(40 * t), "det": -3.0 + 50 * t, "phase": 0.4 * np.ones(T)})
    prob["eigenbasis"] = ["r", "g", "h"]
    prob["basis_name"] = "all"
    prob["samples"]["Local"] = {"digital": {q: {"amp": rng.uniform(2, 8) * np.ones(T), "det": rng.uniform(-3, 3) * np.ones(T),
                                                 "phase": rng.uniform(0, 1) * np.ones(T)} for q in (0, n // 2, n - 2)}}
    init = np.zeros(3**n, dtype=complex)
    init[sum(1 * 3**k for k in range(n))] = 1.0  # |g...g>
    return prob, init, (T - 1) * 1e-3


def xy_problem(n=12):
    """The reference's mesolve-XY test sequence (tests/golden/noisy_xy_2.npz: inputs captured from pulser-core) on a 12-atom
    2 x 6 register at the same 4-um pitch: global XY channel, magnetic field (0, 0, 30)."""
    from pulser_amd import QutipEmulator
    from pulser_amd.hamiltonian_data import SequenceInputs

    prob, _ = load_fixture("noisy_xy_2.npz")
    inp = dict(prob["inputs"])
    inp["coords"] = P.register_coords(P.square_rect(2, n // 2), 4.0)
    inp["qubit_ids"] = tuple(f"atom{k}" for k in range(n))
    ch = dict(inp["channels"][0])
    ch["slots"] = [np.array([int(s[0]), int(s[1])] + list(range(n)), dtype=np.int64) for s in ch["slots"]]
    inp["channels"] = [ch]
    emu = QutipEmulator(SequenceInputs.from_dict(inp), sampling_rate=1.0)
    p = emu._current_problem
    return p, np.asarray(emu.initial_state).reshape(-1), (int(p["duration"]) - 1) * 1e-3
