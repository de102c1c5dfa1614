"""CPU: the numerical claims behind two scheduling decisions of the device stepper, on small NumPy models
(no oracle, no GPU): (1) CF4 steps that span several spline knots on smooth stretches of the anneal stay far
inside the 1e-7 bar (host_sched.hpp: multi-knot steps); (2) the 4th-order 6-stage split-operator composition
with one sub-step per knot does too (k_split.hpp)."""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

from bigstep_probe import cf4_run  # noqa: E402
from ket_split_probe import Prob, reference, split_run  # noqa: E402


def test_multi_knot_cf4_steps_on_the_anneal():
    pr = Prob(2, 2, "rect")  # 4 atoms at the blockade radius, the 3.1-us anneal
    t_end = 1400  # ramp-up, the first kink (with its guard zone of one-knot steps) and 900 ns of the sweep
    ref, _ = reference(pr, t_end, h=0.25, order=22)
    kinks = [0, 500, 2100, 3100]
    errs = {}
    for k in (1, 2, 3, 4):
        psi, steps = cf4_run(pr, t_end, k, kinks)
        errs[k] = np.abs(psi - ref).max()
        assert steps <= t_end / k + 2 * 60  # merged everywhere but in the guard zones
    assert errs[1] < 1e-9
    assert errs[2] < 5e-9 and errs[3] < 2e-8  # what the schedule admits on this pulse: 2 - 3 knots
    assert errs[4] < 1e-7 and errs[2] < errs[3] < errs[4]  # 4th order in the step: grows ~ k^4


def test_split_operator_composition_one_sub_step_per_knot():
    pr = Prob(2, 3, "rect")
    t_end = 1000
    ref, _ = reference(pr, t_end, h=0.25, order=22)
    err = {}
    for name in ("strang", "s6_4"):
        psi, _ = split_run(pr, t_end, 1.0, name)
        err[name] = np.abs(psi - ref).max()
    assert err["s6_4"] < 5e-9  # the device default (k_split.hpp)
    assert err["strang"] > 1e3 * err["s6_4"]  # a 2nd-order split would not do at this step
