import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, warnings
from helpers import blockade_radius
from test_host_logic import _inputs_from_problem
from pulser_amd import QutipEmulator, problem as P
n = int(sys.argv[1])
coords = P.register_coords(P.triangular_rect(2, (n + 1) // 2), blockade_radius())[:n]
prob = P.make_ising_problem(coords, P.anneal_samples())
res = {}
for name in ("windows", "sequential"):
    if name == "sequential": os.environ["PULSER_AMD_NO_WINDOWS"] = "1"
    for rep in range(2):
        emu = QutipEmulator(_inputs_from_problem(prob, "ground-rydberg"), evaluation_times="Full")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = emu.run()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = emu.last_engine_stats
    idx = list(range(0, len(r.states), 97)) + [len(r.states) - 1, 500, 501, 1300]
    res[name] = (np.stack([np.asarray(r.states[i])[:, 0] for i in idx]), st)
    print(name, f"{dt*1e3:.1f} ms", {k: st[k] for k in ("n_applications", "n_launches")}, st["reserved"][0], st.get("windows"))
print("max gap", np.max(np.abs(res["windows"][0] - res["sequential"][0])))
if os.environ.get("WIN_PROFILE"):
    import cProfile, pstats, io
    os.environ.pop("PULSER_AMD_NO_WINDOWS", None)
    emu = QutipEmulator(_inputs_from_problem(prob, "ground-rydberg"), evaluation_times="Full")
    pr = cProfile.Profile(); pr.enable()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = emu.run()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
