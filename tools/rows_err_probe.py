"""Dev probe: error of the split-operator row path (run_rows) against the tight-oracle fixture of the interacting
10-atom triangular register, by block length (split_steps = CF4 steps per half block)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import load_fixture, sketch_errors, with_anneal_samples
from pulser_amd.engine import Engine

prob, extra = load_fixture("cfg3_tri10_dephasing.npz")
prob = with_anneal_samples(prob)
times = np.asarray(extra["eval_times"])
for label, path, kw in (("rows default", dict(force_ket=True), {}), ("rows no_merge", dict(force_ket=True, no_merge=True), {}),
                        ("rows split_steps=1", dict(force_ket=True), {"split_steps": 1}),
                        ("rows split_steps=1 no_merge", dict(force_ket=True, no_merge=True), {"split_steps": 1}),
                        ("rows tol 1e-13", dict(force_ket=True, no_merge=True), {"tol": 1e-13}),
                        ("multi-launch", dict(no_ket=True), {})):
    with Engine.from_problems([prob], mode="mesolve") as eng:
        eng.set_path(False, **path)
        snaps = eng.solve(eng.new_state(), times, **kw).cpu().numpy()[:, 0]
        st = eng.stats()
    errs = [sketch_errors(snaps[k - 1], extra, k) for k in range(1, len(times))]
    print(f"{label:32s} rows-err " + " ".join(f"{e['rows']:.1e}" for e in errs) + "  probes " + " ".join(f"{e['probes']:.1e}" for e in errs)
          + f"  steps {st['n_steps']} stages {st['n_applications']} launches {st['n_launches']}", flush=True)

# the same register without dissipation: the ket kernel against CF4 + Taylor at 1e-13
from pulser_amd import problem as P
from helpers import blockade_radius
coords = P.register_coords(P.triangular_rect(2, 5), blockade_radius())
kprob = P.make_ising_problem(coords, P.anneal_samples())
with Engine.from_problems([kprob], mode="sesolve") as eng:
    eng.set_path(True, no_ket=True, no_merge=True)
    ref = eng.solve(eng.new_state(), times, tol=1e-13).cpu().numpy()[:, 0]
for label, path, kw in (("ket default", dict(force_ket=True), {}), ("ket no_merge", dict(force_ket=True, no_merge=True), {}),
                        ("ket tol 1e-13 no_merge", dict(force_ket=True, no_merge=True), {"tol": 1e-13}),
                        ("k_traj default", dict(), {}), ("taylor 1e-10 merge", dict(no_ket=True), {})):
    with Engine.from_problems([kprob], mode="sesolve") as eng:
        eng.set_path(label.startswith("taylor"), **path)
        snaps = eng.solve(eng.new_state(), times, **kw).cpu().numpy()[:, 0]
        st = eng.stats()
    print(f"{label:32s} |psi - taylor(1e-13)| " + " ".join(f"{np.max(np.abs(snaps[k] - ref[k])):.1e}" for k in range(len(times) - 1))
          + f"  steps {st['n_steps']} stages {st['n_applications']} launches {st['n_launches']}", flush=True)
