#!/bin/bash
# per-pass time of the split-operator pass kernels over variant builds (build/variants/*.so): tools/split_bench.py N, fixed sub-steps
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r4
export RYD_DEV=1
for f in build/variants/*.so; do echo "== $f"; RYD_LIB=$f python tools/split_bench.py ${@:-20} 2>&1 | grep "fixed=True"; done
