"""Errors of the split-operator master equation against the 12-atom tight fixture (tests/golden/cfg3_tri12_dephasing.npz) at
every stored time: default row path (k_split_reg rows, four-knot halves) and the k_ket rows of round 3."""
import os, sys
os.environ.setdefault("RYD_DEV", "1")  # the RYD_* A/B switches this tool reads are ignored without it (dev_common.hpp)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import load_fixture, with_anneal_samples, sketch_errors
from pulser_amd.engine import Engine

prob, extra = load_fixture("cfg3_tri12_dephasing.npz")
prob = with_anneal_samples(prob)
times = np.asarray(extra["eval_times"])
for name, kw in (("default (k_split_reg rows)", {}), ("k_ket rows", {"rows_ket": True})):
    with Engine.from_problems([prob], mode="mesolve") as eng:
        eng.set_path(False, **kw)
        snaps = eng.solve(eng.new_state(), times)
        st = eng.stats()
        comps = [sketch_errors(snaps[k - 1, 0].cpu().numpy(), extra, k) for k in range(1, len(times))]
    print(f"{name} (RYD_ROWS_KH={os.environ.get('RYD_ROWS_KH', '-')}): launches {st['n_launches']}, stages {st['n_applications']}", flush=True)
    for key in ("rows", "diag", "probes", "purity"):
        print(f"   {key:7s} at t = {times[1:]}: {' '.join('%.1e' % c[key] for c in comps)}", flush=True)
