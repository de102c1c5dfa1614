"""Where does the single-launch plan stop paying? (dev probe)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quick_bench import run

if __name__ == "__main__":
    for n, t1 in ((20, 0.052), (21, 0.032), (22, 0.022)):
        for fs in (False, True):
            run(n, "sesolve", t1, force_single=fs)
    for fs in (False, True):
        run(16, "sesolve", 0.032, batch=16, force_single=fs)
        run(17, "sesolve", 0.032, batch=16, force_single=fs)
        run(10, "mesolve", 0.012, force_single=fs)
        run(11, "mesolve", 0.006, force_single=fs)
