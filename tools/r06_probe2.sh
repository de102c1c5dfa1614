cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export RYD_DEV=1
{
for s in 2685 2570 2327; do
  echo "== seed $s default"; python tools/fuzz_one.py $s 2>&1 | grep "default:"
  echo "== seed $s BANK=0"; RYD_SPLIT_BANK=0 python tools/fuzz_one.py $s 2>&1 | grep "default:"
  echo "== seed $s DIV=16"; RYD_SPLIT_PERIOD_DIV=16 python tools/fuzz_one.py $s 2>&1 | grep "default:"
  echo "== seed $s DIV=16 BANK=0"; RYD_SPLIT_BANK=0 RYD_SPLIT_PERIOD_DIV=16 python tools/fuzz_one.py $s 2>&1 | grep "default:"
done
echo "== path 2685"; python tools/fuzz_path.py 2685 8 2>&1 | grep -v amdgpu
echo "== trace 2685 t=150"; RYD_SPLIT_TRACE=1 python - << 'PY' 2>&1 | grep -v amdgpu | cut -c1-200
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import fuzz_case
from pulser_amd.engine import Engine
probs, desc = fuzz_case(2685)
with Engine.from_problems(probs, mode="sesolve") as eng:
    st = eng.new_state(); eng.evolve(st, 0.0, 0.183)
PY
} > gpurun_out/r06_probe2.log 2>&1
cat gpurun_out/r06_probe2.log
