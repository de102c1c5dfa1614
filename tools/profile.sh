#!/bin/bash
# Round profile recipe (run on the GPU box via gpurun): kernel-trace stats for the headline and
# the cfg3 / cfg5 / cfg2 workloads, then FETCH_SIZE / WRITE_SIZE in their own passes
# (MI355X_MICROARCH.md: separate --pmc passes; FETCH_SIZE x2 on gfx950).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*"; timeout 900 "$@" > $OUT/$name.log 2>&1; echo "rc=$?"; tail -2 $OUT/$name.log | cut -c1-300; }
NS="python bench.py --no-cpu --no-extras --no-legs"
run ns_stats rocprofv3 --kernel-trace --stats -d $OUT/ns_stats -o ns --output-format csv -- $NS --steps 2 --warmup 1
run nst_stats rocprofv3 --kernel-trace --stats -d $OUT/nst_stats -o nst --output-format csv -- $NS --split-turns --steps 2 --warmup 1
run nsk_stats rocprofv3 --kernel-trace --stats -d $OUT/nsk_stats -o nsk --output-format csv -- $NS --no-split14 --steps 2 --warmup 1
run cfg3_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg3_stats -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 8
run cfg5_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg5_stats -o cfg5 --output-format csv -- python bench.py --workload cfg5 --steps 2 --warmup 1 --slice-ns 50
run cfg5t_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg5t_stats -o cfg5t --output-format csv -- python bench.py --workload cfg5 --method taylor --steps 2 --warmup 1 --slice-ns 20
run cfg5b_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg5b_stats -o cfg5b --output-format csv -- python bench.py --workload cfg5 --atoms 24 --steps 2 --warmup 1 --slice-ns 10
run cfg5c_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg5c_stats -o cfg5c --output-format csv -- python bench.py --workload cfg5 --atoms 22 --steps 2 --warmup 1 --slice-ns 10
run cfg2_stats rocprofv3 --kernel-trace --stats -d $OUT/cfg2_stats -o cfg2 --output-format csv -- python bench.py --workload cfg2 --steps 2 --warmup 1
run ns_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/ns_fetch -o ns --output-format csv -- $NS --steps 1 --warmup 0
run ns_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/ns_write -o ns --output-format csv -- $NS --steps 1 --warmup 0
run nst_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/nst_fetch -o nst --output-format csv -- $NS --split-turns --steps 1 --warmup 0
run nst_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/nst_write -o nst --output-format csv -- $NS --split-turns --steps 1 --warmup 0
run nsk_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/nsk_fetch -o nsk --output-format csv -- $NS --no-split14 --steps 1 --warmup 0
run nsk_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/nsk_write -o nsk --output-format csv -- $NS --no-split14 --steps 1 --warmup 0
run cfg3_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg3_fetch -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 4
run cfg3_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg3_write -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 4
run cfg5_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg5_fetch -o cfg5 --output-format csv -- python bench.py --workload cfg5 --steps 1 --warmup 0 --slice-ns 8
run cfg5_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg5_write -o cfg5 --output-format csv -- python bench.py --workload cfg5 --steps 1 --warmup 0 --slice-ns 8
run cfg5t_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg5t_fetch -o cfg5t --output-format csv -- python bench.py --workload cfg5 --method taylor --steps 1 --warmup 0 --slice-ns 4
run cfg5t_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg5t_write -o cfg5t --output-format csv -- python bench.py --workload cfg5 --method taylor --steps 1 --warmup 0 --slice-ns 4
run cfg5b_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg5b_fetch -o cfg5b --output-format csv -- python bench.py --workload cfg5 --atoms 24 --steps 1 --warmup 0 --slice-ns 4
run cfg5b_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg5b_write -o cfg5b --output-format csv -- python bench.py --workload cfg5 --atoms 24 --steps 1 --warmup 0 --slice-ns 4
run cfg5c_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg5c_fetch -o cfg5c --output-format csv -- python bench.py --workload cfg5 --atoms 22 --steps 1 --warmup 0 --slice-ns 4
run cfg5c_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg5c_write -o cfg5c --output-format csv -- python bench.py --workload cfg5 --atoms 22 --steps 1 --warmup 0 --slice-ns 4
run cfg2_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg2_fetch -o cfg2 --output-format csv -- python bench.py --workload cfg2 --steps 1 --warmup 0
run cfg2_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg2_write -o cfg2 --output-format csv -- python bench.py --workload cfg2 --steps 1 --warmup 0
find $OUT -type f | head -60
# keep only small files for the merge back
find $OUT -type f -size +4M -delete
