"""Dev probe (round 4): step-size controller of the split-operator path on the headline register: stages, estimate and TRUE
error (tight-oracle fixture, six times) for a batch of 8 identical sequences.  RYD_SPLIT_GROW = hysteresis of growth."""
import os, sys
os.environ.setdefault("RYD_DEV", "1")  # the RYD_* A/B switches this tool reads are ignored without it (dev_common.hpp)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import load_fixture, with_anneal_samples
from pulser_amd.engine import Engine

prob, extra = load_fixture("ns_tri14_anneal.npz")
prob = with_anneal_samples(prob)
times = np.asarray(extra["eval_times"]); ref = np.asarray(extra["oracle_states_tight"])
for B in (8, 1):
    with Engine.from_problems([prob] * B, mode="sesolve") as eng:
        snaps = eng.solve(eng.new_state(), times).cpu().numpy()
        st = eng.stats()
    errs = [float(np.max(np.abs(snaps[k - 1][0] - ref[k]))) for k in range(1, len(ref))]
    print(f"GROW={os.environ.get('RYD_SPLIT_GROW', '1.6')} B={B}: stages {st['n_applications']}, launches {st['n_launches']}, estimate {st['reserved'][0]:.2e}, "
          f"restores {st['reserved'][3]:.0f}, true errors {' '.join('%.1e' % e for e in errs)}", flush=True)
with Engine.from_problems([prob] * 8, mode="sesolve") as eng:
    st0 = eng.new_state(); eng.evolve(st0, 0.0, 3.1); s = eng.stats()
    print(f"   whole anneal in one call: stages {s['n_applications']}, estimate {s['reserved'][0]:.2e}, error {np.max(np.abs(st0.cpu().numpy()[0] - ref[-1])):.1e}")
