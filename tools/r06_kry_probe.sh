#!/bin/bash
# round 6: the fused Lanczos iteration (k_apply epilogue reductions + k_kry_update_fused) against the five-launch one
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_ket.py tests/test_gpu_fullsize.py -q -m gpu -k "krylov or cfg5 or 16" 2>&1 | tail -4
echo "== fused"; python bench.py --workload cfg5 --method krylov --steps 1 --warmup 1 --slice-ns 20 2>&1 | tail -1 | cut -c1-700
echo "== RYD_KRY_FUSE=0"; RYD_DEV=1 RYD_KRY_FUSE=0 python bench.py --workload cfg5 --method krylov --steps 1 --warmup 1 --slice-ns 20 2>&1 | tail -1 | cut -c1-700
python - << 'PY'
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from helpers import blockade_radius
from pulser_amd import problem as P
from pulser_amd.engine import Engine
coords = P.register_coords(P.square_rect(4, 5), blockade_radius())
prob = P.make_ising_problem(coords, P.anneal_samples())
outs = {}
for name, env in (("fused", "1"), ("plain", "0")):
    pass
with Engine.from_problems([prob], mode="sesolve") as eng:
    a = eng.new_state(); eng.evolve(a, 1.0, 1.05, method="krylov", tol=1e-12)
    b = eng.new_state(); eng.evolve(b, 1.0, 1.05, method="taylor", tol=1e-12)
    print("20 atoms, 50 ns at 1 us from the ground state: |krylov - taylor| =", float((a - b).abs().max()), "norm - 1 =", float(torch.linalg.vector_norm(a)) - 1.0)
PY
} > gpurun_out/r06_kry_probe.log 2>&1
grep -v amdgpu gpurun_out/r06_kry_probe.log
