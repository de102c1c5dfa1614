#!/bin/bash
# Round profile recipe (run on the GPU box via gpurun): kernel-trace stats for the
# primary and the HBM-streaming workloads, then FETCH_SIZE / WRITE_SIZE in their
# own passes (MI355X_MICROARCH.md: separate --pmc passes; FETCH_SIZE x2 on gfx950).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*"; timeout 600 "$@" > $OUT/$name.log 2>&1; echo "rc=$?"; tail -2 $OUT/$name.log | cut -c1-400; }
run cfg2_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg2_stats -o cfg2 --output-format csv -- python bench.py --no-cpu --no-extras --steps 2 --warmup 1
run cfg3_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg3_stats -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 2 --warmup 1
run cfg5_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg5_stats -o cfg5 --output-format csv -- python bench.py --workload cfg5 --steps 2 --warmup 1 --slice-ns 20
run cfg3_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg3_fetch -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 1
run cfg3_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg3_write -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 1
run cfg2_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg2_fetch -o cfg2 --output-format csv -- python bench.py --no-cpu --no-extras --steps 1 --warmup 0
run cfg2_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg2_write -o cfg2 --output-format csv -- python bench.py --no-cpu --no-extras --steps 1 --warmup 0
find $OUT -type f | head -50
# keep only small files for the merge back
find $OUT -type f -size +4M -delete
