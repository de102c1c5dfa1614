"""Mid-size / large-state timing probe of the tiled multi-launch kernels (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quick_bench import run

if __name__ == "__main__":
    run(16, "sesolve", 0.052)
    run(18, "sesolve", 0.052)
    run(20, "sesolve", 0.052)
    run(22, "sesolve", 0.012)
    run(24, "sesolve", 0.007)
    run(26, "sesolve", 0.004)
    run(16, "sesolve", 0.022, batch=64)
    run(10, "mesolve", 0.052)
    run(12, "mesolve", 0.012)
    run(13, "mesolve", 0.004)
