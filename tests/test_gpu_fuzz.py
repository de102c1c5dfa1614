"""-m gpu: seeded fuzz of the split-operator step-size controller.

The reference's integrator is adaptive per step (simulation.py:768-780: zvode with atol / rtol / max_step); the default
path of 12+-atom kets here is a split-operator composition whose sub-step is MEASURED (one sub-step against its two
halves) at intervals and booked - a heuristic that has to be attacked with inputs nobody hand-picked.  Sequences are drawn
from the waveform families Pulser ships (pulser-core/pulser/waveforms.py: Constant / square incl. back-to-back pulses with
phase jumps and EOM-style 1-ns edges, Ramp, Blackman, Kaiser, Interpolated (PCHIP), Composite; delays), on chains,
two-row triangular and rectangular registers of 12, 13, 14 and 16 atoms at spacings of 4.5 - 10 um (nearest-neighbour
interactions of 5 - 650 rad/us), durations of 100 - 4 000 ns, batches of 1 - 4 DIFFERENT sequences (tests/helpers.py:
fuzz_case).  Every case: the default path against CF4 + Taylor at tol 1e-12 (an a-priori tolerance) - the amplitudes
within the stated bar (1e-7) AND within 4 x the estimate the controller booked (ryd_stats.reserved[0], what the
Python engine compares with the budget and warns about).

The named regressions are the cases the fuzz found while the controller was being fixed in round 5 (docs/KERNEL_NOTES.md 5.10 (vi)):
seed 40  - 1 240 one-knot steps of a strongly interacting chain ran unchecked (8.7e-8, estimate 2.1e-8);
seed 263 - a square pulse from t = 0 on 16 atoms: the one check of the sequence measured the product state (1.85e-7 / 2.8e-8);
seed 279 - the same on one-knot steps of the 6-stage scheme (4.8e-9 / 7.8e-10);
seed 306, 72 - 4.5-um chains (7.3e-8 / 1.5e-8, 4.3e-8 / 2.8e-8);
seed 1197 - found by a second sweep (seeds 400 - 1199): the 6th-order kind's sub-step stood at 18 ns from two checks at zero
           amplitude, every later step of that kind was 2 - 3 knots long and "uninformative" against it (8.1e-7 / 1.8e-9); and a
           roll-back re-measured at the tame checkpoint and grew the sub-step back to what had just failed.

Round 6, second hold-out (seeds 2000 - 2999; DESIGN 5.10, step-size control; profiles/r06_fuzz_summary.md):
seed 2685 - ramp / plateau at 24 rad/us / ramp on a 13-atom chain, 183 ns: 1.19e-7 with an estimate of 2.0e-8.  The controller
           had measured the LARGEST ENTRY of the local error; the state was spread over thousands of basis states (largest entry
           of the error 1.4e-9 of a 2-norm of 4.9e-8), and the falling ramp gathered population and error back into a few of them.
           The controller measures the 2-norm of the local error now: the booked estimate BOUNDS the final error.
seeds 2570, 2327 - error 5.2 x / 4.0 x the largest-entry estimate (3.5e-8, 4.7e-8).
"""
from __future__ import annotations

import warnings

import numpy as np
import pytest

from helpers import SPLIT_BUDGET, fuzz_case

pytestmark = pytest.mark.gpu

AMP_TOL = 1e-7
COVER = 4.0          # error <= COVER x booked estimate ...
FLOOR = 2e-9         # ... above this floor (below it the reference's own 1e-10 .. 1e-9 and rounding take over)


def _run_case(seed):
    from pulser_amd.engine import Engine

    probs, desc = fuzz_case(seed)
    t_end = (probs[0]["duration"] - 1) * 1e-3
    with Engine.from_problems(probs, mode="sesolve") as eng:
        ref = eng.new_state()
        eng.evolve(ref, 0.0, t_end, method="taylor", tol=1e-12, magnus_tol=1e-12)
        st = eng.new_state()
        eng.reset_stats()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)  # (a budget overrun is asserted on below, not warned about)
            eng.evolve(st, 0.0, t_end)
        s = eng.stats()
        err = float((st - ref).abs().max())
    return err, s["reserved"][0], s, desc


@pytest.mark.parametrize("block", range(8))
def test_controller_fuzz_default_path_against_a_priori_tolerance(block):
    """80 seeded cases in blocks of 10 (< 40 s together on one MI355X)."""
    worst = 0.0
    for seed in range(10 * block, 10 * block + 10):
        err, est, s, desc = _run_case(seed)
        assert err < AMP_TOL, (desc, err, est)
        if est > 0.0:  # the split-operator path ran under its controller
            assert err <= max(COVER * est, FLOOR), (desc, err, est, s["reserved"][:4])
            assert est < 2.0 * SPLIT_BUDGET, (desc, est)  # the budget of a sequence (the engine warns beyond 2 x)
            if err > FLOOR:
                worst = max(worst, err / est)
    print(f"block {block}: worst error / estimate above the floor = {worst:.2f}")


@pytest.mark.parametrize("seed", [40, 72, 92, 129, 137, 177, 263, 265, 279, 306, 359, 599, 1187, 1197])
def test_controller_fuzz_named_regressions(seed):
    err, est, s, desc = _run_case(seed)
    print(f"{desc}: error {err:.2e}, estimate {est:.2e}, stages {s['n_applications']}, roll-backs {s['reserved'][3]:.0f}")
    assert err < AMP_TOL / 2, (desc, err, est)
    assert est > 0.0 and err <= max(COVER * est, FLOOR) and est < 2.0 * SPLIT_BUDGET, (desc, err, est)


@pytest.mark.parametrize("seed", [2685, 2570, 2327, 2244, 2799, 985])
def test_controller_estimate_bounds_the_error_second_holdout(seed):
    """The cases the second hold-out flagged under the largest-entry controller (2685: 1.19e-7), the hold-out's worst error under
    the 2-norm controller (2799: 3.2e-8) and the worst error / estimate of all 3 000 seeds (985: 1.2).  The estimate is a sum of
    local 2-norms: it has to COVER the error (x 1.5 for the sparsity of the measurements), not only come within a factor of 4."""
    err, est, s, desc = _run_case(seed)
    print(f"{desc}: error {err:.2e}, estimate {est:.2e}, stages {s['n_applications']}, roll-backs {s['reserved'][3]:.0f}")
    assert err < AMP_TOL / 2, (desc, err, est)
    assert est > 0.0 and err <= max(1.5 * est, FLOOR) and est <= 1.25 * SPLIT_BUDGET, (desc, err, est)


# ---------------------------------------------------------------------------------------------------------------------
# Oracle-pinned cases: the comparisons above are HIP against HIP (two algorithm families of this repo).
# tests/golden/make_fuzz_fixtures.py integrated a subset of the same seeded cases with the tight CPU oracle (zvode rtol
# 1e-13: the restatement of the reference's solver call, simulation.py:729-735, 768-780); here the default path AND the
# CF4 + Taylor path that serves as the fuzz reference are both held to those kets.
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_cases(name):
    import os

    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_fuzz_fixtures import digest

    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    for k, seed in enumerate(fx["seeds"]):
        over = int(fx["n_override"][k])
        probs, desc = fuzz_case(int(seed), None if over < 0 else over)
        rows = [r for r in range(len(fx["state_owner"])) if fx["state_owner"][r][0] == k]
        refs = []
        for r in rows:
            b = int(fx["state_owner"][r][1])
            assert digest(probs[b]) == str(fx["input_sha256"][r]), f"fuzz_case({seed}) has drifted from the fixture"
            refs.append((b, fx["states"][r][: 2 ** int(fx["state_atoms"][r])]))
        yield probs, desc, refs


@pytest.mark.parametrize("name", ["fuzz_oracle_12.npz", "fuzz_oracle_13.npz", "fuzz_oracle_14.npz"])
def test_default_path_and_taylor_reference_against_the_tight_oracle(name):
    """24 / 8 / 8 fuzz seeds of 12 / 13 / 14 atoms (the first of each size, not picked by outcome): final kets of the
    default path (k_split_reg under the step-size controller) and of CF4 + Taylor at tol 1e-12, both within 1e-7 of the
    tight oracle; the controller's booked estimate covers the default path's true error (x 4 above the 2e-9 floor)."""
    from pulser_amd.engine import Engine

    worst = [0.0, 0.0]
    n_cases = 0
    for probs, desc, refs in _oracle_cases(name):
        t_end = (probs[0]["duration"] - 1) * 1e-3
        with Engine.from_problems(probs, mode="sesolve") as eng:
            tay = eng.new_state()
            eng.evolve(tay, 0.0, t_end, method="taylor", tol=1e-12, magnus_tol=1e-12)
            st = eng.new_state()
            eng.reset_stats()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                eng.evolve(st, 0.0, t_end)
            est = eng.stats()["reserved"][0]
            got, ref_t = st.cpu().numpy(), tay.cpu().numpy()
        for b, ref in refs:
            e_def = float(np.max(np.abs(got[b] - ref)))
            e_tay = float(np.max(np.abs(ref_t[b] - ref)))
            worst = [max(worst[0], e_def), max(worst[1], e_tay)]
            assert e_tay < 1e-8, (desc, b, e_tay)          # the fuzz's reference path is itself oracle-pinned
            assert e_def < AMP_TOL, (desc, b, e_def)
            assert e_def <= max(COVER * est, FLOOR), (desc, b, e_def, est)
        n_cases += 1
    assert n_cases >= 8
    print(f"{name}: {n_cases} cases, worst |default - oracle| {worst[0]:.2e}, worst |taylor - oracle| {worst[1]:.2e}")


def test_second_holdout_cases_against_the_tight_oracle():
    """The cases the second hold-out flagged under the largest-entry controller (seed 2685: 1.19e-7 then) and the worst cases of
    the 2-norm controller - by error, by error / estimate, by 2-norm of the error / estimate - against the TIGHT ORACLE
    (tests/golden/fuzz_oracle_holdout.npz, make_fuzz_fixtures.py holdout: 12 - 16 atoms, ten cases): default path and CF4 + Taylor
    within the bar, and the booked estimate - a sum of local 2-norms - covers the 2-norm of the true error (measured <= 1.9 x)."""
    from pulser_amd.engine import Engine

    n_cases = 0
    for probs, desc, refs in _oracle_cases("fuzz_oracle_holdout.npz"):
        t_end = (probs[0]["duration"] - 1) * 1e-3
        with Engine.from_problems(probs, mode="sesolve") as eng:
            tay = eng.new_state()
            eng.evolve(tay, 0.0, t_end, method="taylor", tol=1e-12, magnus_tol=1e-12)
            st = eng.new_state()
            eng.reset_stats()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                eng.evolve(st, 0.0, t_end)
            est = eng.stats()["reserved"][0]
            got, ref_t = st.cpu().numpy(), tay.cpu().numpy()
        for b, ref in refs:
            e_def = float(np.max(np.abs(got[b] - ref)))
            e_two = float(np.linalg.norm(got[b] - ref))
            e_tay = float(np.max(np.abs(ref_t[b] - ref)))
            print(f"{desc}: |default - oracle| max {e_def:.2e} 2-norm {e_two:.2e}, estimate {est:.2e}; |taylor - oracle| {e_tay:.2e}")
            # (the oracle shows that the largest "errors" the fuzz books on strongly interacting registers are the CF4 + Taylor
            # REFERENCE's - seed 2799, a 4.56-um chain with a time-dependent phase: 3.2e-8; 2745, 2013: 1e-8 - its a-priori
            # Magnus estimate is short there at tol 1e-12, while the default path sits 6e-11 .. 5e-10 from the oracle:
            # profiles/r06_fuzz_summary.md, last table)
            assert e_tay < 1e-8, (desc, b, e_tay)  # (with the interaction-strength rule of host_sched.hpp; before it: 3.2e-8 on seed 2799)
            assert e_def < AMP_TOL / 2, (desc, b, e_def)
            assert e_two <= max(2.5 * est, FLOOR), (desc, b, e_two, est)  # the estimate covers the 2-NORM of the error (measured: <= 1.9 x)
        n_cases += 1
    assert n_cases == 10


def test_strongly_interacting_registers_picked_by_input_against_the_tight_oracle():
    """16 fuzz seeds of 12 / 13 atoms at 4.5 - 5.5 um picked by their INPUT (tests/golden/fuzz_oracle_strong.npz: none of them was
    used to fit the interaction-strength rule of host_sched.hpp): the default path within the bar with its estimate covering the
    2-norm of the error, and the CF4 + Taylor reference (tol = magnus_tol = 1e-12) within 1e-8 of the tight oracle."""
    from pulser_amd.engine import Engine

    worst = [0.0, 0.0, 0.0]
    n_cases = 0
    for probs, desc, refs in _oracle_cases("fuzz_oracle_strong.npz"):
        t_end = (probs[0]["duration"] - 1) * 1e-3
        with Engine.from_problems(probs, mode="sesolve") as eng:
            tay = eng.new_state()
            eng.evolve(tay, 0.0, t_end, method="taylor", tol=1e-12, magnus_tol=1e-12)
            st = eng.new_state()
            eng.reset_stats()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                eng.evolve(st, 0.0, t_end)
            est = eng.stats()["reserved"][0]
            got, ref_t = st.cpu().numpy(), tay.cpu().numpy()
        for b, ref in refs:
            e_def = float(np.max(np.abs(got[b] - ref)))
            e_two = float(np.linalg.norm(got[b] - ref))
            e_tay = float(np.max(np.abs(ref_t[b] - ref)))
            worst = [max(worst[0], e_def), max(worst[1], e_tay), max(worst[2], e_two / est if e_two > FLOOR else 0.0)]
            print(f"{desc}: |default - oracle| max {e_def:.2e} 2-norm {e_two:.2e}, estimate {est:.2e}; |taylor - oracle| {e_tay:.2e}")
            assert e_def < AMP_TOL / 2, (desc, b, e_def)
            assert e_two <= max(2.5 * est, FLOOR), (desc, b, e_two, est)
            assert e_tay < 1e-8, (desc, b, e_tay)
        n_cases += 1
    assert n_cases == 16
    print(f"worst |default - oracle| {worst[0]:.2e}, worst |taylor - oracle| {worst[1]:.2e}, worst 2-norm / estimate {worst[2]:.2f}")


def test_default_path_of_small_strongly_interacting_registers_against_the_tight_oracle():
    """8 - 11 atoms at 4.5 - 5.4 um (300 - 650 rad/us between neighbours): the DEFAULT path there is the persistent polynomial
    kernel (k_traj: CF4 steps from a-priori estimates).  Round 6 found those estimates short on such registers at 12 - 16 atoms
    (CF4 + Taylor 3.2e-8 from the oracle at tol 1e-12) and added an interaction-strength rule to host_sched.hpp; here the default
    call (default tolerances) and the tight Taylor call are held to the oracle on the small sizes."""
    from pulser_amd.engine import Engine

    n_cases = 0
    for probs, desc, refs in _oracle_cases("fuzz_oracle_strong_small.npz"):
        t_end = (probs[0]["duration"] - 1) * 1e-3
        with Engine.from_problems(probs[:1], mode="sesolve") as eng:
            st = eng.new_state()
            eng.evolve(st, 0.0, t_end)
            tay = eng.new_state()
            eng.evolve(tay, 0.0, t_end, method="taylor", tol=1e-12, magnus_tol=1e-12)
            got, ref_t = st.cpu().numpy(), tay.cpu().numpy()
        for b, ref in refs:
            e_def = float(np.max(np.abs(got[b] - ref)))
            e_tay = float(np.max(np.abs(ref_t[b] - ref)))
            print(f"{desc}: |default - oracle| {e_def:.2e}, |taylor(1e-12) - oracle| {e_tay:.2e}")
            assert e_def < AMP_TOL, (desc, e_def)
            assert e_tay < 1e-8, (desc, e_tay)
        n_cases += 1
    assert n_cases == 5


def test_split_path_forced_on_small_registers_against_the_tight_oracle():
    """24 seeds re-drawn on 8 - 11 atoms with the split-operator path forced (method = "split": the pass kernels under the
    same controller), against the tight oracle: <= 1e-7 and covered by the estimate."""
    from pulser_amd.engine import Engine

    n_cases = 0
    for probs, desc, refs in _oracle_cases("fuzz_oracle_small.npz"):
        t_end = (probs[0]["duration"] - 1) * 1e-3
        with Engine.from_problems(probs, mode="sesolve") as eng:
            st = eng.new_state()
            eng.reset_stats()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                eng.evolve(st, 0.0, t_end, method="split")
            s = eng.stats()
            got = st.cpu().numpy()
        assert s["reserved"][0] > 0.0, desc  # the controller ran (the split path was taken)
        for b, ref in refs:
            e = float(np.max(np.abs(got[b] - ref)))
            assert e < AMP_TOL and e <= max(COVER * s["reserved"][0], FLOOR), (desc, b, e, s["reserved"][0])
        n_cases += 1
    assert n_cases == 24
