#!/bin/bash
# round 6: the cost weight of the one-knot (kind 1) steps under the 2-norm controller: headline stages and parity by weight
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export RYD_DEV=1
{
for w in 4 6 8 12; do
  echo "== W1=$w"; RYD_SPLIT_W1=$w python bench.py --no-cpu --no-legs --no-extras 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stages_per_sequence'], d['config']['parity_max_abs'])"
done
echo "== trace W1=8"; RYD_SPLIT_W1=8 RYD_SPLIT_TRACE=1 python tools/trace_ctrl.py 14 minimal 2>&1 | grep -v amdgpu | cut -c1-250 | grep "kind 1\|n_applications"
} > gpurun_out/r06_w1_probe.log 2>&1
cat gpurun_out/r06_w1_probe.log
