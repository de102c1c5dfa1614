"""Lanczos(m) vs CF4 + Taylor on the same slice: applications, launches and wall-clock per
simulated us at 16 and 20 atoms (profiles/r02_krylov_vs_taylor.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

def rect(rows, cols):
    coords = P.register_coords(P.square_rect(rows, cols), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples())

print("| N | method | tol | slice | applications/us | launches/us | order or m | ms per sim-ns | sim-us/s | algorithmic GB per sim-us |")
print("|---|---|---|---|---|---|---|---|---|---|")
for n, (r, c), span in ((16, (4, 4), 0.05), (20, (4, 5), 0.02)):
    prob = rect(r, c)
    ref = None
    for method, tol in (("taylor", 1e-10), ("krylov", 1e-10), ("taylor", 1e-12), ("krylov", 1e-12)):
        with Engine.from_problems([prob], mode="sesolve") as eng:
            st = eng.new_state(); eng.evolve(st, 0.0, 1.0 if False else 0.005); torch.cuda.synchronize()
            st = eng.new_state(); eng.evolve(st, 0.0, 0.005, method=method, tol=tol); torch.cuda.synchronize()
            st = eng.new_state(); eng.reset_stats()
            t0 = 1.0
            tic = time.perf_counter(); eng.evolve(st, t0, t0 + span, method=method, tol=tol); torch.cuda.synchronize()
            dt = time.perf_counter() - tic
            s = eng.stats()
            apps = s["n_applications"] / span
            # Taylor stage: read w, read base, write out = 48 B/amp; Lanczos vector: apply 32 + dot 32 + update 48 (+32 with v_{j-1}) + normalize 32 B/amp, combine 16 m + 16
            print(f"| {n} | {method} | {tol:g} | {span*1e3:.0f} ns at t = 1 us | {apps:.0f} | {s['n_launches']/span:.0f} | {s['last_order']} | {dt/span/1e3*1e3:.3f} | {span/dt:.3f} | {32.0*2**n*apps/1e9:.1f} |", flush=True)
