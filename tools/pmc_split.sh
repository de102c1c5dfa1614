#!/bin/bash
# SQ counters of the split-operator pass kernel k_split_s (20-atom ket, 20 ns slice = ~125 passes)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_split; rm -rf $OUT; mkdir -p $OUT
SET1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SET2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
SET3="SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH GRBM_GUI_ACTIVE SQ_INSTS_FLAT"
i=0
for S in "$SET1" "$SET2" "$SET3"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $S --kernel-trace -d $OUT/p$i -o c --output-format csv -- python bench.py --workload cfg5 --steps 1 --warmup 0 --slice-ns 20 > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob
for i in (1,2,3):
    f=glob.glob(f"gpurun_out/pmc_split/p{i}/c_counter_collection.csv")
    if not f: print("pass", i, "no output"); continue
    rows=[r for r in csv.DictReader(open(f[0])) if 'k_split_s' in r['Kernel_Name']]
    acc={}
    for r in rows: acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    print({k: (len(v), round(sum(v)/1e6,3)) for k,v in acc.items()})
    if rows: print("launches", len(rows)//max(len(acc),1), "VGPR", rows[0]['VGPR_Count'], "SGPR", rows[0]['SGPR_Count'], "LDS", rows[0]['LDS_Block_Size'], "scratch", rows[0]['Scratch_Size'])
PY
find $OUT -type f -size +2M -delete
