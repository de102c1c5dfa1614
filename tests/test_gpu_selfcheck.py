"""-m gpu: the library's self-check (norm drift, NaN scan), enabled with RYD_CHECK=1.
The switch is read once per process, so the checked runs happen in a child process."""
from __future__ import annotations

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from helpers import chain_problem
from pulser_amd._lib import RydError
from pulser_amd.engine import Engine

for n, mode, ops in ((8, "sesolve", None), (5, "mesolve", [(np.sqrt(0.1), "sigma_rr")]), (15, "sesolve", None)):
    with Engine.from_problems([chain_problem(n, ops)], mode=mode) as eng:
        st = eng.new_state()
        eng.evolve(st, 0.0, 0.05)          # healthy run: passes the check
        bad = eng.new_state()
        bad[0].view(-1)[3] = float("nan")
        try:
            eng.evolve(bad, 0.0, 0.01)
            print("MISSED", n, mode)
        except RydError as exc:
            assert exc.code == -5 and "non-finite" in str(exc), exc
            print("CAUGHT", n, mode)
# a stepper driven outside its stability region: Taylor order 2 on a strong drive loses the norm
with Engine.from_problems([chain_problem(8)], mode="sesolve") as eng:
    eng.set_path(True)
    st = eng.new_state()
    try:
        eng.evolve(st, 0.6, 1.6, taylor_order=2, max_step=0.02)
        print("MISSED drift")
    except RydError as exc:
        assert exc.code == -5 and "squared norm" in str(exc), exc
        print("CAUGHT drift")
"""


def test_selfcheck_catches_nan_and_norm_drift_and_passes_healthy_runs():
    env = dict(os.environ, RYD_CHECK="1")
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith(("CAUGHT", "MISSED"))]
    assert lines == ["CAUGHT 8 sesolve", "CAUGHT 5 mesolve", "CAUGHT 15 sesolve", "CAUGHT drift"], out.stdout


def test_selfcheck_is_off_by_default():
    env = {k: v for k, v in os.environ.items() if k != "RYD_CHECK"}
    code = CHILD.format(root=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "CAUGHT" not in out.stdout  # NaNs flow through silently without the switch (no hidden syncs)
