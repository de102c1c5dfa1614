#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/pmc3; rm -rf $OUT; mkdir -p $OUT
SET1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SET2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"
for variant in fast slow; do
  [ $variant = slow ] && export RYD_NO_FAST_APPLY=1 || unset RYD_NO_FAST_APPLY
  i=0
  for S in "$SET1" "$SET2"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $S --kernel-trace -d $OUT/${variant}_$i -o c --output-format csv -- python bench.py --workload cfg3 --steps 1 --warmup 0 --slice-ns 1 > $OUT/${variant}_$i.log 2>&1
  done
done
python - <<'PY'
import csv, glob
from collections import defaultdict
for variant in ("fast","slow"):
    acc=defaultdict(lambda: defaultdict(list))
    for i in (1,2):
        f=glob.glob(f"gpurun_out/pmc3/{variant}_{i}/c_counter_collection.csv")
        if not f: print("missing", variant, i); continue
        rows=list(csv.DictReader(open(f[0])))
        # group per dispatch to know pass index
        disp=defaultdict(dict)
        for r in rows:
            if 'k_apply' not in r['Kernel_Name']: continue
            disp[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
        ids=sorted(disp)
        for n,d in enumerate(ids):
            for c,v in disp[d].items(): acc[n%3][c].append(v)
    for p in range(3):
        print(variant,"pass",p, {c: round(sum(v)/len(v)/1e6,2) for c,v in sorted(acc[p].items())})
PY
find $OUT -type f -size +2M -delete
