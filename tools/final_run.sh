#!/bin/bash
# End-of-round evidence run (gpurun): smoke, full GPU test suite, bench lines.  Profiles: tools/profile.sh.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
R=${1:-r4}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1; echo "smoke rc=$?"
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${R}_pytest_final.log; tail -2 gpurun_out/${R}_pytest_final.log
python bench.py > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.err; echo "bench rc=$?"
python bench.py --workload cfg4 --steps 2 --warmup 1 > gpurun_out/${R}_bench_cfg4.json 2>/dev/null; echo "cfg4 rc=$?"
python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu > gpurun_out/${R}_bench_cfg2.json 2>/dev/null; echo "cfg2 rc=$?"
