"""Where does the single-launch plan stop paying? (dev probe)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quick_bench import run

if __name__ == "__main__":
    # density matrices: Hermitian path (2^14 register-tile row pass + symmetrisation) vs one launch
    run(11, "mesolve", 0.006)
    run(11, "mesolve", 0.006, no14=True)
    run(10, "mesolve", 0.012)
    run(10, "mesolve", 0.012, force14=True)
    run(9, "mesolve", 0.012, batch=16)
    run(9, "mesolve", 0.012, batch=16, no14=True)
    # defaults after the policy change
    run(21, "sesolve", 0.032)
    run(15, "sesolve", 0.012, batch=256)
    run(14, "sesolve", 0.022, batch=256)
