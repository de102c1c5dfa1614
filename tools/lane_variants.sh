#!/bin/bash
# timing of the variant builds of k_split_reg (build/variants/*.so: hipcc -DSPLITR_...), fixed stage count;
# RYD_SPLIT_NR picks the shape (5: 512 lanes x 32 amplitudes, 6: 256 x 64)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r4
export RYD_SPLIT_NR=${RYD_SPLIT_NR:-5} RYD_DEV=1
for f in build/variants/*.so; do RYD_LIB=$f python tools/lane_check.py 2>&1 | grep -v amdgpu.ids | head -1; RYD_LIB=$f python tools/lane_time.py 256; done
