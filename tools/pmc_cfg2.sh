#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc2; rm -rf $OUT; mkdir -p $OUT
SET1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SET2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
i=0
for S in "$SET1" "$SET2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $S --kernel-trace -d $OUT/p$i -o c --output-format csv -- python bench.py --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob
for i in (1,2):
    f=glob.glob(f"gpurun_out/pmc2/p{i}/c_counter_collection.csv")
    rows=[r for r in csv.DictReader(open(f[0])) if 'k_traj' in r['Kernel_Name']]
    print({r['Counter_Name']: round(float(r['Counter_Value'])/1e6,1) for r in rows})
    if rows: print("VGPR", rows[0]['VGPR_Count'], "SGPR", rows[0]['SGPR_Count'], "LDS", rows[0]['LDS_Block_Size'], "scratch", rows[0]['Scratch_Size'])
PY
find $OUT -type f -size +2M -delete
