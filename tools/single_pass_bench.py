"""Single-launch plan (partner tiles through L2 / Infinity Cache) vs the multi-pass tiling (dev probe):
single kets of 14-17 atoms, small batches, 7-8-atom density matrices."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quick_bench import run

if __name__ == "__main__":
    for n, t1 in ((14, 0.202), (15, 0.152), (16, 0.102), (17, 0.082)):
        for no_single in (True, False):
            run(n, "sesolve", t1, no_single=no_single)
    for no_single in (True, False):
        run(14, "sesolve", 0.102, batch=8, no_single=no_single)
        run(7, "mesolve", 0.102, no_single=no_single)
        run(8, "mesolve", 0.052, no_single=no_single)
