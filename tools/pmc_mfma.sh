#!/bin/bash
# PMC pass for the fp64 MFMA aggregation kernel (run on the GPU box via gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_mfma
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES --kernel-trace -d $OUT/a -o mfma --output-format csv -- python tools/outer_bench.py > $OUT/a.log 2>&1
echo rc=$?
tail -3 $OUT/a.log
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob("gpurun_out/prof_mfma/a/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "k_outer_mfma" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]]["sum"] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k, v in acc.items():
    print(k, v["sum"], "over", n[k], "dispatch rows")
PY
find $OUT -type f -size +2M -delete
