"""GPU check of the split-operator ket path (k_split): errors against the tight oracle fixtures and
the Taylor path at several sizes / tilings, and timings.  python tools/split_check.py [quick]"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from helpers import blockade_radius, load_fixture, with_anneal_samples  # noqa: E402
from pulser_amd import problem as P  # noqa: E402
from pulser_amd.engine import Engine  # noqa: E402


def rect(rows, cols):
    coords = P.register_coords(P.square_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples())


def local_problem(n, seed=0, duration=61, spacing=7.0, phase=True):
    rng = np.random.default_rng(seed)
    t = np.arange(duration) / 1000.0
    coords = P.register_coords(P.square_rect(1, n), spacing) + rng.normal(0, 0.3, (n, 2))
    z = np.zeros(duration)
    prob = P.make_ising_problem(coords, {"amp": z, "det": z, "phase": z})
    loc = {}
    for q in range(n):
        a, b, c = rng.uniform(2, 12, 3)
        loc[q] = {"amp": a * (1 + 0.5 * np.sin(2 * np.pi * (q + 1) * t / t[-1])),
                  "det": b * np.cos(3 * t + q) - c,
                  "phase": (0.3 * q + 2.0 * t) if phase else z}
    prob["samples"] = {"Global": {}, "Local": {"ground-rydberg": loc}}
    return prob


def stats_line(s):
    return (f"stages={s['n_applications']} launches={s['n_launches']} steps={s['n_steps']} "
            f"est_err={s['reserved'][0]:.2e} last_e={s['reserved'][1]:.2e} tau={s['reserved'][2]:.2e} "
            f"rollbacks={s['reserved'][3]:.0f}")


def fixture_case(name):
    prob, extra = load_fixture(name)
    prob = with_anneal_samples(prob)
    times = np.asarray(extra["eval_times"])
    ref = np.asarray(extra["oracle_states_tight"])
    for fixed in (True, False):
        with Engine.from_problems([prob], mode="sesolve") as eng:
            eng.set_path(False, split_fixed=fixed)
            t0 = time.time()
            snaps = eng.solve(eng.new_state(), times, method="split").cpu().numpy()[:, 0]
            dt = time.time() - t0
            s = eng.stats()
        errs = [np.max(np.abs(snaps[k - 1] - ref[k])) for k in range(1, len(times))]
        print(f"{name} fixed={fixed}: max err vs tight oracle {max(errs):.3e} (per snapshot {['%.1e' % e for e in errs]}) "
              f"{dt:.2f}s {stats_line(s)}", flush=True)


def versus_taylor(label, prob, t_a, t_b, batch=1, **kw):
    outs = {}
    times = {}
    st_line = ""
    for method in ("taylor", "split"):
        probs = [prob] * batch
        with Engine.from_problems(probs, mode="sesolve") as eng:
            st = eng.new_state()
            if t_a > 0:
                eng.evolve(st, 0.0, t_a, method="taylor", tol=1e-12)
            torch.cuda.synchronize()
            t0 = time.time()
            if method == "taylor":
                eng.evolve(st, t_a, t_b, method="taylor", tol=1e-12)
            else:
                eng.evolve(st, t_a, t_b, method="split", **kw)
            torch.cuda.synchronize()
            times[method] = time.time() - t0
            outs[method] = st.cpu().numpy()
            if method == "split":
                st_line = stats_line(eng.stats())
    d = np.max(np.abs(outs["taylor"] - outs["split"]))
    nrm = abs(np.linalg.norm(outs["split"][0]) - 1.0)
    print(f"{label}: |split - taylor|max = {d:.3e}  norm-1 = {nrm:.1e}  taylor {times['taylor'] * 1e3:.1f} ms  "
          f"split {times['split'] * 1e3:.1f} ms  ({(t_b - t_a) * 1e3:.0f} ns)  {st_line}", flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    fixture_case("cfg2_chain8_anneal.npz")
    fixture_case("cfg2_chain12_anneal.npz")
    versus_taylor("local complex drives, 10 atoms, 60 ns", local_problem(10), 0.0, 0.06)
    versus_taylor("local complex drives, 13 atoms (2 tilings), 60 ns", local_problem(13, seed=3), 0.0, 0.06)
    versus_taylor("rect 2x7 = 14 atoms, 100 ns at 1 us", rect(2, 7), 1.0, 1.1)
    versus_taylor("rect 4x4 = 16 atoms, 100 ns at 1 us", rect(4, 4), 1.0, 1.1)
    versus_taylor("rect 4x4 = 16 atoms x 3 sequences", rect(4, 4), 1.0, 1.05, batch=3)
    versus_taylor("rect 4x5 = 20 atoms, 50 ns at 1 us", rect(4, 5), 1.0, 1.05)
    if not quick:
        versus_taylor("rect 2x11 = 22 atoms (3 tilings), 10 ns at 1 us", rect(2, 11), 1.0, 1.01)
        # cfg5 throughput: 500 ns of the 20-atom anneal with the controller
        prob = rect(4, 5)
        with Engine.from_problems([prob], mode="sesolve") as eng:
            st = eng.new_state()
            eng.evolve(st, 0.0, 0.01, method="split")
            torch.cuda.synchronize()
            t0 = time.time()
            eng.evolve(st, 0.01, 0.51, method="split")
            torch.cuda.synchronize()
            dt = time.time() - t0
            print(f"20 atoms, 500 ns from t = 10 ns: {dt * 1e3:.1f} ms -> {0.5 / dt:.2f} sim-us/s  {stats_line(eng.stats())}",
                  flush=True)
            eng.set_path(False, split_fixed=True)
            torch.cuda.synchronize()
            t0 = time.time()
            eng.evolve(st, 0.51, 1.01, method="split")
            torch.cuda.synchronize()
            dt = time.time() - t0
            print(f"20 atoms, 500 ns, no controller: {dt * 1e3:.1f} ms -> {0.5 / dt:.2f} sim-us/s", flush=True)


if __name__ == "__main__":
    main()
