#!/bin/bash
# End-of-round evidence run (gpurun): smoke, full GPU test suite, bench lines, counters.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo "smoke rc=$?"
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r2_pytest_final.log; tail -2 gpurun_out/r2_pytest_final.log
python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; echo "bench rc=$?"
python bench.py --no-extras --no-cpu --full-lindblad --steps 1 --warmup 0 > gpurun_out/r2_bench_full_lindblad.json 2> gpurun_out/r2_bench_full_lindblad.err; echo "full lindblad rc=$?"
python bench.py --workload cfg4 --steps 2 --warmup 1 > gpurun_out/r2_bench_cfg4.json 2>&1; echo "cfg4 rc=$?"
bash tools/pmc_ket.sh > gpurun_out/r2_pmc_ket.log 2>&1; tail -6 gpurun_out/r2_pmc_ket.log
