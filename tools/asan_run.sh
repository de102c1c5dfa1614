#!/bin/bash
# AddressSanitizer run of the host side of librydemu (make -C pulser_amd/csrc asan builds build/librydemu_asan.so).
#   tools/asan_run.sh cpu   the C-ABI tests that need no GPU (symbols, ABI version, argument validation, loud failures)
#   tools/asan_run.sh gpu   (on the GPU box, via gpurun) the host-heavy GPU tests: schedule planner and merge logic,
#                           the split-operator controller with its roll-backs, run_rows, quantum jumps, ensembles.
#                           MEASURED round 4: with the ASan runtime preloaded the ROCm runtime of this image aborts inside
#                           its own initialisation (torch.cuda.is_available(): ROCr's address-space reservation and
#                           ASan's shadow do not coexist without the instrumented ROCm libraries + xnack), so only
#                           the cpu mode is usable here; the target is kept for images that ship the ASan ROCm stack.
# ASan reports abort the run (abort_on_error); leak detection is off (the Python interpreter never frees everything).
cd "$(dirname "$0")/.."
[ -f build/librydemu_asan.so ] || make -C pulser_amd/csrc asan || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0 RYD_DEV=1 RYD_LIB=$PWD/build/librydemu_asan.so
if [ "${1:-cpu}" = cpu ]; then
  python -m pytest tests/test_host_logic.py -q -m "not gpu" -k "symbol or abi or null or header or loudly"
else
  python -m pytest tests/test_gpu_split.py tests/test_gpu_merge.py tests/test_gpu_mcwf.py tests/test_gpu_distributed.py -q -m gpu -x \
    && python -m pytest tests/test_gpu_ket.py -q -m gpu -x -k "rows or gauge or pulse" \
    && python -m pytest tests/test_gpu_emulator.py -q -m gpu -x -k "noisy or cfg4"
fi
