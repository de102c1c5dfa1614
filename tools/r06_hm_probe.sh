#!/bin/bash
# round 6: half-mirror lane basis (SPLITR_HM) and scheduler-flag variants of k_split_reg<14, 5>; traces of the fuzz seeds
# the second hold-out (2000 - 2999) flagged
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export RYD_DEV=1
{
python tools/lane_check.py 2>&1 | grep -v amdgpu.ids | head -3
python tools/lane_time.py 256
python tools/lane_time.py 256
for f in build/variants/*.so; do RYD_LIB=$f python tools/lane_check.py 2>&1 | grep -v amdgpu.ids | head -1; RYD_LIB=$f python tools/lane_time.py 256; done
} > gpurun_out/r06_hm_variants.log 2>&1
for s in 2685 2570 2244; do
  RYD_SPLIT_TRACE=1 python tools/fuzz_one.py $s > gpurun_out/r06_fuzz_seed$s.log 2>&1
  python tools/fuzz_locate.py $s 8 > gpurun_out/r06_fuzz_locate$s.log 2>&1
done
python -m pytest tests/test_gpu_split.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r06_hm_pytest.log 2>&1
tail -3 gpurun_out/r06_hm_pytest.log; cat gpurun_out/r06_hm_variants.log
