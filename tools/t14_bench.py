import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from quick_bench import run
for B in (16, 64, 128, 256, 512):
    t1 = 0.002 + 0.05 * min(1.0, 64 / B)
    run(14, "sesolve", t1, batch=B)
    run(14, "sesolve", t1, batch=B, no14=True)
for n in (16, 18):
    for B in (4, 16, 64):
        run(n, "sesolve", 0.012, batch=B)
        run(n, "sesolve", 0.012, batch=B, force14=True)
