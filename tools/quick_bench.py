"""Ad-hoc timing probe (dev tool, not the judged bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import chain_problem
from pulser_amd.engine import Engine

def run(n, mode, t1, batch=1, order=0, tol=0.0, tile_bits=0, no14=False, force14=False, no_single=False, force_single=False):
    ops = [(np.sqrt(0.1), "sigma_rr")] if mode == "mesolve" else None
    prob = chain_problem(n, collapse_ops=ops)
    eng = Engine.from_problems([prob] * batch, mode=mode, tile_bits=tile_bits)
    if no14 or force14 or no_single or force_single:
        eng.set_path(False, no_tile14=no14, force_tile14=force14, no_single_pass=no_single,
                     force_single_pass=force_single)
    st = eng.new_state()
    eng.evolve(st, 0.0, 0.002, taylor_order=order, tol=tol)
    torch.cuda.synchronize()
    eng.reset_stats()
    t0 = time.time()
    eng.evolve(st, 0.002, t1, taylor_order=order, tol=tol)
    torch.cuda.synchronize()
    dt = time.time() - t0
    s = eng.stats()
    nb = n if mode == "sesolve" else 2 * n
    bytes_alg = 32.0 * (2 ** nb) * batch * s["n_applications"]
    tag = (" no14" if no14 else "") + (" force14" if force14 else "") + (" multi-pass" if no_single else "") + (" forced-single" if force_single else "")
    print(f"N={n} {mode} B={batch} tile={tile_bits or 12}{tag}: {t1-0.002:.3f} us in {dt*1e3:.1f} ms -> {(t1-0.002)*batch/dt:.3f} sim-us/s; "
          f"apps {s['n_applications']} launches {s['n_launches']} order {s['last_order']} bound {s['norm_bound']:.0f}; "
          f"{dt/s['n_launches']*1e6:.2f} us/launch; alg BW {bytes_alg/dt/1e12:.3f} TB/s", flush=True)
    eng.close()

if __name__ == "__main__":
    run(12, "sesolve", 3.1)
    run(12, "sesolve", 0.502, batch=256)
    run(12, "sesolve", 0.302, batch=1024)
    run(10, "sesolve", 3.1)
    run(8, "sesolve", 3.1)
    sys.exit(0)
    run(12, "sesolve", 0.302)
    run(12, "sesolve", 0.102, batch=256)
    run(12, "sesolve", 0.052, batch=1024)
    run(14, "sesolve", 0.302)
    run(20, "sesolve", 0.052)
    run(20, "sesolve", 0.052, tile_bits=11)
    run(24, "sesolve", 0.007)
    run(10, "mesolve", 0.052)
    run(12, "mesolve", 0.012)
    run(13, "mesolve", 0.004)
    run(14, "mesolve", 0.003)
