"""Dev probe: register-resident ket kernel and split-operator rows at 14 atoms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine

def tri(ops=None):
    coords = P.register_coords(P.triangular_rect(2, 7), blockade_radius())
    return P.make_ising_problem(coords, P.anneal_samples(), collapse_ops=ops)

def run(eng, t0, t1, **kw):
    st = eng.new_state(); torch.cuda.synchronize(); eng.reset_stats()
    tic = time.perf_counter(); eng.evolve(st, t0, t1, **kw); torch.cuda.synchronize()
    return time.perf_counter() - tic, eng.stats(), st

which = sys.argv[1:] or ["single", "batch", "rows"]
if "single" in which:
    for no_ket in (False, True):
        with Engine.from_problems([tri()], mode="sesolve") as eng:
            eng.set_path(False, no_ket=no_ket)
            run(eng, 0.0, 0.01)
            t1 = 3.1
            dt, s, _ = run(eng, 0.0, t1)
            print(f"single 14-atom, no_ket={no_ket}: {t1/dt:.2f} sim-us/s ({dt:.3f} s for {t1} us), apps/ns {s['n_applications']/(t1*1e3):.1f}, stats {s}", flush=True)
if "batch" in which:
    for B in (256, 1024):
        with Engine.from_problems([tri()] * B, mode="sesolve") as eng:
            run(eng, 0.0, 0.01)
            dt, s, _ = run(eng, 1.0, 1.1)
            print(f"batch {B} x 14-atom: {B*0.1/dt:.1f} sim-us/s ({dt*1e3:.1f} ms per 100 ns), per stage-row {dt/ (s['n_applications'])*1e6*min(B,256)/B:.2f} us, stats {s}", flush=True)
if "rows" in which:
    ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr")]
    with Engine.from_problems([tri(ops)], mode="mesolve") as eng:
        run(eng, 1.0, 1.004)
        for K in (4, 8):
            dt, s, st = run(eng, 1.0, 1.016, split_steps=K)
            tr = float(torch.diagonal(st[0]).real.sum().item())
            print(f"cfg3 rows K={K}: {0.016/dt:.4f} sim-us/s ({dt/16*1e3:.2f} ms per ns), trace {tr:.15f}, stats {s}", flush=True)
        eng.set_path(False, no_ket=True)
        dt, s, st = run(eng, 1.0, 1.002)
        print(f"cfg3 hermitian path: {0.002/dt:.4f} sim-us/s ({dt/2*1e3:.2f} ms per ns), stats {s}", flush=True)

if "pmc" in which:
    with Engine.from_problems([tri()] * 256, mode="sesolve") as eng:
        dt, s, _ = run(eng, 1.0, 1.02)
        print(f"pmc leg: 256 x 14-atom, 20 ns: {dt*1e3:.1f} ms, stats {s}", flush=True)

if "relax" in which:
    ops = [(float(np.sqrt(2 * 0.05)), "sigma_rr"), (float(np.sqrt(0.02)), "sigma_gr")]
    with Engine.from_problems([tri(ops)], mode="mesolve") as eng:
        run(eng, 1.0, 1.002)
        dt, s, st = run(eng, 1.0, 1.008)
        tr = float(torch.diagonal(st[0]).real.sum().item())
        print(f"14 atoms dephasing + relaxation, rows: {dt/8*1e3:.2f} ms per ns, trace {tr:.15f}, stats {s}", flush=True)
        eng.set_path(False, no_ket=True)
        dt, s, st = run(eng, 1.0, 1.001)
        print(f"14 atoms dephasing + relaxation, multi-launch pair passes: {dt/1*1e3:.2f} ms per ns, stats {s}", flush=True)
