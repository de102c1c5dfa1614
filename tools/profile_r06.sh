#!/bin/bash
# Round-6 profile recipe (gpurun): kernel-trace stats for the headline, cfg2, cfg3, cfg5 and the drop-in call with
# evaluation_times="Full", then FETCH_SIZE / WRITE_SIZE in their own passes (MI355X_MICROARCH.md: separate --pmc
# passes; FETCH_SIZE x2 on gfx950).  Summarise with: python tools/summarize_prof.py gpurun_out/prof r06
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*"; timeout 600 "$@" > $OUT/$name.log 2>&1; echo "rc=$?"; tail -1 $OUT/$name.log | cut -c1-200; }
NS="python bench.py --no-cpu --no-extras --no-legs"
run ns_stats rocprofv3 --kernel-trace --stats -d $OUT/ns_stats -o ns --output-format csv -- $NS --steps 2 --warmup 1
run cfg2_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg2_stats -o cfg2 --output-format csv -- python bench.py --workload cfg2 --no-extras --no-cpu --steps 2 --warmup 1
run cfg3_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg3_stats -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 8
run cfg5_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg5_stats -o cfg5 --output-format csv -- python bench.py --workload cfg5 --steps 2 --warmup 1 --slice-ns 50
run cfg5k_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg5k_stats -o cfg5k --output-format csv -- python bench.py --workload cfg5 --method krylov --steps 1 --warmup 1 --slice-ns 10
run api_stats rocprofv3 --kernel-trace --stats -d $OUT/api_stats -o api --output-format csv -- python tools/api_bench.py --atoms 14 --repeat 1
run ns_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/ns_fetch -o ns --output-format csv -- $NS --steps 1 --warmup 0
run ns_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/ns_write -o ns --output-format csv -- $NS --steps 1 --warmup 0
run cfg2_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg2_fetch -o cfg2 --output-format csv -- python bench.py --workload cfg2 --no-extras --no-cpu --steps 1 --warmup 0
run cfg2_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg2_write -o cfg2 --output-format csv -- python bench.py --workload cfg2 --no-extras --no-cpu --steps 1 --warmup 0
run cfg3_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg3_fetch -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 8
run cfg3_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg3_write -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 8
run cfg5_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg5_fetch -o cfg5 --output-format csv -- python bench.py --workload cfg5 --steps 1 --warmup 0 --slice-ns 8
run cfg5_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg5_write -o cfg5 --output-format csv -- python bench.py --workload cfg5 --steps 1 --warmup 0 --slice-ns 8
run cfg5b_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg5b_fetch -o cfg5b --output-format csv -- python bench.py --workload cfg5 --atoms 24 --steps 1 --warmup 0 --slice-ns 4
run cfg5b_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg5b_write -o cfg5b --output-format csv -- python bench.py --workload cfg5 --atoms 24 --steps 1 --warmup 0 --slice-ns 4
# keep only small files for the merge back
find $OUT -type f -size +4M -delete
find $OUT -type f | wc -l
