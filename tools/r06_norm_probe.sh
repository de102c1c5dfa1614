#!/bin/bash
# round 6: the controller on the 2-norm of the difference (RYD_SPLIT_NORM=2, default) against the largest entry (0):
# checks of the headline anneal with the ratio of the two norms, stage counts, the flagged fuzz seeds, the bench headline
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export RYD_DEV=1
{
echo "== trace 14 atoms minimal, NORM=2"; RYD_SPLIT_TRACE=1 python tools/trace_ctrl.py 14 minimal 2>&1 | grep -v amdgpu | cut -c1-260
echo "== trace 14 atoms minimal, NORM=0"; RYD_SPLIT_NORM=0 python tools/trace_ctrl.py 14 minimal 2>&1 | grep -v amdgpu | tail -1
for s in 2685 2570 2327 2244 263 1197; do
  echo "== seed $s NORM=2"; python tools/fuzz_one.py $s 2>&1 | grep "default:"
done
echo "== bench NORM=2"; python bench.py --no-cpu --no-legs --no-extras 2>&1 | tail -1 | cut -c1-900
echo "== bench NORM=0"; RYD_SPLIT_NORM=0 python bench.py --no-cpu --no-legs --no-extras 2>&1 | tail -1 | cut -c1-900
} > gpurun_out/r06_norm_probe.log 2>&1
cat gpurun_out/r06_norm_probe.log
