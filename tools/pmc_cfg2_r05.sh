#!/bin/bash
# SQ counters of k_split_reg<12, 4> on the cfg2 batch (256 x 12 atoms, full anneal, one step): three --pmc passes
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_cfg2; rm -rf $OUT; mkdir -p $OUT
SET1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SET2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
i=0
for S in "$SET1" "$SET2"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $S --kernel-trace -d $OUT/p$i -o c --output-format csv -- python bench.py --workload cfg2 --no-extras --no-cpu --steps 1 --warmup 0 > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob
for i in (1, 2):
    f = glob.glob(f"gpurun_out/pmc_cfg2/p{i}/c_counter_collection.csv")
    if not f: print("pass", i, "no output"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "k_split_reg<12, 4" in r["Kernel_Name"]]
    acc = {}
    for r in rows: acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print({k: (len(v), round(sum(v) / 1e6, 3)) for k, v in acc.items()})
    if rows: print("launches", len(rows) // max(len(acc), 1), "VGPR", rows[0]["VGPR_Count"], "SGPR", rows[0]["SGPR_Count"], "LDS", rows[0]["LDS_Block_Size"], "scratch", rows[0]["Scratch_Size"])
PY
find $OUT -type f -size +2M -delete
