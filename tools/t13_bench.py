import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from quick_bench import run
for B in (1, 16, 256):
    t1 = 0.002 + 0.05 * min(1.0, 32 / B)
    run(13, "sesolve", t1, batch=B)
    run(13, "sesolve", t1, batch=B, tile_bits=13)
run(21, "sesolve", 0.012)
run(21, "sesolve", 0.012, tile_bits=13)
run(22, "sesolve", 0.012)
run(22, "sesolve", 0.012, tile_bits=13)
run(15, "sesolve", 0.012, batch=64)
run(15, "sesolve", 0.012, batch=64, tile_bits=13)
