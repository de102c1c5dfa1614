#!/bin/bash
# round 6: the flagged seeds of the second fuzz hold-out on the controller with the period cap, the GPU tests that run on
# k_split_reg, then the hold-out itself again
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
export RYD_DEV=1
for s in 2685 2570 2244; do
  RYD_SPLIT_TRACE=1 python tools/fuzz_one.py $s 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_fuzz_seed${s}_cap.log
  tail -4 gpurun_out/r06_fuzz_seed${s}_cap.log | head -2
done
python -m pytest tests/test_gpu_split.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_ket.py -x -q -m gpu > gpurun_out/r06_hm2_pytest.log 2>&1
tail -3 gpurun_out/r06_hm2_pytest.log
python tools/fuzz_ctrl.py 2000 1000 > gpurun_out/r06_fuzz_cap_2000_1000.log 2>&1; tail -12 gpurun_out/r06_fuzz_cap_2000_1000.log
