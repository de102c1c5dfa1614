"""-m gpu: multi-knot CF4 steps (host_sched.hpp).  Where every waveform is the same polynomial across
spline knots (linear ramps, plateaus) a step spans 2 - 4 knot intervals: fewer exponentials of higher degree.
Checked against the tight oracle over the whole anneal, against the one-knot schedule, and on waveforms
where no knot may be removed."""
from __future__ import annotations

import numpy as np
import pytest

from helpers import blockade_radius, load_fixture, with_anneal_samples
from pulser_amd import problem as P

pytestmark = pytest.mark.gpu


def _engine(probs, mode="sesolve"):
    from pulser_amd.engine import Engine

    return Engine.from_problems(probs, mode=mode)


@pytest.mark.parametrize("fixture", ["cfg2_chain8_anneal.npz", "cfg2_chain12_anneal.npz"])
def test_multi_knot_steps_keep_the_anneal_inside_the_bar(fixture):
    prob, extra = load_fixture(fixture)
    prob = with_anneal_samples(prob)
    times = np.asarray(extra["eval_times"])
    ref = np.asarray(extra["oracle_states_tight"])
    res = {}
    for no_merge in (False, True):
        with _engine([prob]) as eng:
            eng.set_path(False, no_merge=no_merge)
            snaps = eng.solve(eng.new_state(), times).cpu().numpy()[:, 0]
            res[no_merge] = (max(np.max(np.abs(snaps[k - 1] - ref[k])) for k in range(1, len(times))), eng.stats())
    assert res[False][0] < 2e-8 and res[True][0] < 2e-8  # the bar is 1e-7
    # most of the 3100 knot intervals of the anneal are merged (ramps and plateaus; not the ~25 knots of
    # spline ringing on either side of the three kinks), and the work drops with them
    assert res[False][1]["n_steps"] < 0.7 * res[True][1]["n_steps"]
    assert res[False][1]["n_applications"] < 0.8 * res[True][1]["n_applications"]


def test_no_knot_is_removed_where_the_waveform_is_not_one_polynomial():
    """Sinusoidal per-atom amplitudes: every spline piece differs from its neighbour."""
    n, T = 10, 61
    rng = np.random.default_rng(0)
    t = np.arange(T) / 1000.0
    coords = P.register_coords(P.square_rect(1, n), 7.0)
    z = np.zeros(T)
    prob = P.make_ising_problem(coords, {"amp": z, "det": z, "phase": z})
    prob["samples"] = {"Global": {}, "Local": {"ground-rydberg": {
        q: {"amp": 6.0 * (1 + 0.5 * np.sin(2 * np.pi * (q + 1) * t / t[-1])), "det": 4.0 * np.cos(3 * t + q) - 2.0,
            "phase": z} for q in range(n)}}}
    steps = []
    for no_merge in (False, True):
        with _engine([prob]) as eng:
            eng.set_path(False, no_merge=no_merge)
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.06)
            steps.append((eng.stats()["n_steps"], st.cpu().numpy()))
    assert steps[0][0] == steps[1][0]
    assert np.array_equal(steps[0][1], steps[1][1])


def test_multi_knot_steps_in_the_split_operator_master_equation():
    """10 atoms, dephasing, a slice of the anneal's detuning sweep: two-knot steps (one per half block of the
    4th-order splitting) against the one-knot schedule."""
    n = 10
    coords = P.register_coords(P.triangular_rect(2, 5), blockade_radius())
    prob = P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=[(float(np.sqrt(2 * 0.05)), "sigma_rr")])
    out = {}
    with _engine([prob], "mesolve") as eng:
        start = eng.new_state()
        eng.evolve(start, 0.0, 0.9)
        for no_merge in (False, True):
            eng.set_path(False, no_merge=no_merge)
            st = start.clone()
            eng.reset_stats()
            eng.evolve(st, 0.9, 1.0)
            out[no_merge] = (st.cpu().numpy()[0], eng.stats())
    assert out[False][1]["n_steps"] <= 0.55 * out[True][1]["n_steps"]
    assert np.max(np.abs(out[False][0] - out[True][0])) < 2e-8
    assert abs(np.trace(out[False][0]).real - 1.0) < 1e-9


def long_anneal_problem(n, cycles=4):
    """``cycles`` anneals back to back (12.4 us for four): the budgets of the per-exponential defaults grow
    with the number of exponentials, so they are derived from a whole-sequence budget (host_sched.hpp:
    budget_scale) - this is the long-sequence case that checks it."""
    one = P.anneal_samples()
    s = {k: np.concatenate([v[:-1]] * cycles + [v[-1:]]) for k, v in one.items()}
    coords = P.register_coords(P.square_rect(1, n), blockade_radius())
    return P.make_ising_problem(coords, s)


def test_long_sequence_stays_inside_the_bar_on_every_ket_path():
    """10 atoms, 12.4 us, against the tight oracle integrated here (zvode rtol 1e-13)."""
    from oracle import qutip_path as qp

    n = 10
    prob = long_anneal_problem(n)
    T = (prob["duration"] - 1) * 1e-3
    times = np.array([0.0, 0.5 * T, T])
    opts = dict(qp.default_options([np.stack([prob["samples"]["Global"]["ground-rydberg"]["amp"],
                                              prob["samples"]["Global"]["ground-rydberg"]["det"]])], prob["duration"] - 1))
    opts.update(qp.TIGHT)
    ref = qp.sesolve(qp.build_hamiltonian(prob), qp.all_ground_state(n, prob["eigenbasis"]), times, **opts)
    for path in ({}, {"force_ket": True}, {"force_generic": True}):
        with _engine([prob]) as eng:
            eng.set_path(path.pop("force_generic", False), **path)
            snaps = eng.solve(eng.new_state(), times).cpu().numpy()[:, 0]
        errs = [float(np.max(np.abs(snaps[k - 1] - ref[k]))) for k in (1, 2)]
        assert max(errs) < 1e-7, (path, errs)
