"""Tile size of the single-launch plan: more, smaller tiles (more partner reads) vs fewer, larger ones (dev probe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quick_bench import run

if __name__ == "__main__":
    for n, t1, tiles in ((15, 0.102, (9, 10, 11)), (16, 0.082, (9, 10, 11)), (17, 0.062, (9, 10, 11, 12)),
                         (18, 0.052, (10, 11, 12)), (20, 0.032, (10, 11, 12))):
        for t in tiles:
            run(n, "sesolve", t1, tile_bits=t, force_single=True)
