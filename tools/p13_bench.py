import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import chain_problem
from pulser_amd.engine import Engine
res = {}
for B, t1 in ((1, 0.3), (256, 0.3)):
    eng = Engine.from_problems([chain_problem(13)] * B, mode="sesolve")
    st = eng.new_state(); eng.evolve(st, 0.0, 0.002); torch.cuda.synchronize(); eng.reset_stats()
    st = eng.new_state()
    t0 = time.time(); eng.evolve(st, 0.0, t1); torch.cuda.synchronize(); dt = time.time() - t0
    s = eng.stats()
    print(f"P13={os.environ.get('RYD_P13')} B={B}: {t1*B/dt:.1f} sim-us/s, launches {s['n_launches']}, apps {s['n_applications']}, "
          f"{dt/s['n_applications']*1e6:.2f} us/application", flush=True)
    np.save(f"/tmp/p13_{os.environ.get('RYD_P13','0')}_{B}.npy", st[0].cpu().numpy())
    eng.close()
