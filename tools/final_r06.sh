#!/bin/bash
# End-of-round-6 evidence run (gpurun): smoke, the whole GPU suite, the default bench (driver-format line last), profiles.
# Afterwards: python tools/summarize_prof.py gpurun_out/prof r06; cp the bench line / detail to profiles/.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r06_final3}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"
python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; tail -2 gpurun_out/${T}_pytest.log
python bench.py > gpurun_out/${T}_bench_default.out 2> gpurun_out/${T}_bench_default.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/${T}_bench_detail.json
tail -1 gpurun_out/${T}_bench_default.out | cut -c1-1800
bash tools/profile_r06.sh > gpurun_out/${T}_profile.log 2>&1; tail -1 gpurun_out/${T}_profile.log
