#!/bin/bash
# round 6: the 2-norm controller at budget 8e-8: the whole GPU suite, the headline, the second hold-out of the fuzz
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x > gpurun_out/r06_n2_pytest.log 2>&1; tail -5 gpurun_out/r06_n2_pytest.log
python bench.py --no-cpu --no-legs --no-extras 2>&1 | tail -1 | cut -c1-1000 > gpurun_out/r06_n2_bench_head.log; cat gpurun_out/r06_n2_bench_head.log
python tools/fuzz_ctrl.py 2000 1000 > gpurun_out/r06_fuzz_n2_2000_1000.log 2>&1; tail -12 gpurun_out/r06_fuzz_n2_2000_1000.log
