#!/bin/bash
# SQ counters of the register-resident ket kernel (256 x 14-atom sequences, 100 ns slice)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_ket; rm -rf $OUT; mkdir -p $OUT
SET1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SET2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
SET3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_IFETCH GRBM_GUI_ACTIVE SQ_INSTS_FLAT"
i=0
for S in "$SET1" "$SET2" "$SET3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $S --kernel-trace -d $OUT/p$i -o c --output-format csv -- python tools/ket_bench.py pmc > $OUT/p$i.log 2>&1
done
python - <<'PY'
import csv, glob
for i in (1,2,3):
    f=glob.glob(f"gpurun_out/pmc_ket/p{i}/c_counter_collection.csv")
    if not f: print("pass", i, "no output"); continue
    rows=[r for r in csv.DictReader(open(f[0])) if 'k_ket' in r['Kernel_Name']]
    acc={}
    for r in rows: acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    print({k: [round(x/1e6,2) for x in v] for k,v in acc.items()})
    if rows: print("VGPR", rows[0]['VGPR_Count'], "SGPR", rows[0]['SGPR_Count'], "LDS", rows[0]['LDS_Block_Size'], "scratch", rows[0]['Scratch_Size'])
PY
find $OUT -type f -size +2M -delete
