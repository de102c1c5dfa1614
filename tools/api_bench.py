#!/usr/bin/env python
"""What a drop-in user gets: wall clock of ``QutipEmulator(<north-star inputs>).run()`` with the reference's
default arguments (``evaluation_times="Full"``, simulation.py:137, 961), with "Minimal" and with 0.1 - construction,
lowering, handle + upload, solve, state copies and result objects included.

    python tools/api_bench.py [--atoms 14] [--profile] [--touch]

--touch: read every stored state back (np.asarray of res.states[i]) - what a user who plots an observable pays.
"""
from __future__ import annotations

import argparse
import cProfile
import io
import os
import pstats
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def north_star_inputs(n_atoms: int = 14):
    from pulser_amd import problem as P
    from pulser_amd.hamiltonian_data import single_global_channel

    rb = (P.C6_LEVEL70 / (4 * 2 * np.pi / 2)) ** (1 / 6)
    shape = {14: ("tri", 2, 7), 12: ("chain", 1, 12), 13: ("chain", 1, 13), 10: ("tri", 2, 5), 8: ("tri", 2, 4)}[n_atoms]
    lay = P.triangular_rect(shape[1], shape[2]) if shape[0] == "tri" else P.square_rect(shape[1], shape[2])
    coords = P.register_coords(lay, rb)
    smp = {k: v[:-1] for k, v in P.anneal_samples().items()}
    return single_global_channel(coords, smp, P.C6_LEVEL70, extended=False)


def one_call(inputs, spec, touch=False):
    import torch
    from pulser_amd import QutipEmulator

    torch.cuda.synchronize()
    tic = time.perf_counter()
    emu = QutipEmulator(inputs, evaluation_times=spec)
    t_ctor = time.perf_counter() - tic
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        res = emu.run()
    final = np.asarray(res.states[-1])
    torch.cuda.synchronize()
    t_run = time.perf_counter() - tic - t_ctor
    t_touch = 0.0
    if touch:
        t2 = time.perf_counter()
        acc = 0.0
        for s in res.states:
            acc += float(np.abs(np.asarray(s)[0, 0]))
        t_touch = time.perf_counter() - t2
    return {"ctor_s": t_ctor, "run_s": t_run, "total_s": t_ctor + t_run, "touch_all_s": t_touch,
            "n_eval": len(emu.evaluation_times), "stats": dict(emu.last_engine_stats), "final": final}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--atoms", type=int, default=14)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--touch", action="store_true")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    inputs = north_star_inputs(args.atoms)
    for spec in ("Minimal", "Full", 0.1):
        for rep in range(args.repeat):
            if args.profile and rep == args.repeat - 1:
                pr = cProfile.Profile()
                pr.enable()
            r = one_call(inputs, spec, args.touch)
            if args.profile and rep == args.repeat - 1:
                pr.disable()
                s = io.StringIO()
                pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
                print(s.getvalue()[:6000])
            st = r["stats"]
            print(f"[api] atoms={args.atoms} eval={spec!r:9} rep={rep} n_eval={r['n_eval']:5d} ctor={r['ctor_s']*1e3:8.1f} ms "
                  f"run={r['run_s']*1e3:9.1f} ms total={r['total_s']*1e3:9.1f} ms touch={r['touch_all_s']*1e3:8.1f} ms "
                  f"stages={st.get('n_applications')} launches={st.get('n_launches')} est={st.get('reserved', [0])[0]:.2e} "
                  f"sim-us/s={3.1 / r['total_s']:.2f}", flush=True)


if __name__ == "__main__":
    main()
