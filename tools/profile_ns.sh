#!/bin/bash
# Headline-only profile (after the default kernel of the 14-atom batch changed to k_split14_loop): kernel-trace stats of
# the default bench step and of the k_ket step (--no-split14), FETCH_SIZE / WRITE_SIZE of the default step in their own passes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_ns
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*"; timeout 600 "$@" > $OUT/$name.log 2>&1; echo "rc=$?"; tail -1 $OUT/$name.log | cut -c1-200; }
NS="python bench.py --no-cpu --no-extras --no-legs"
run ns_stats rocprofv3 --kernel-trace --stats -d $OUT/ns_stats -o ns --output-format csv -- $NS --steps 2 --warmup 1
run nsk_stats rocprofv3 --kernel-trace --stats -d $OUT/nsk_stats -o nsk --output-format csv -- $NS --no-split14 --steps 2 --warmup 1
run ns_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/ns_fetch -o ns --output-format csv -- $NS --steps 1 --warmup 0
run ns_write rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/ns_write -o ns --output-format csv -- $NS --steps 1 --warmup 0
find $OUT -type f -size +4M -delete
find $OUT -type f | head -40
