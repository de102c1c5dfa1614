#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof3; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o cfg3 --output-format csv -- python bench.py --workload cfg3 --steps 2 --warmup 1 > $OUT/log.txt 2>&1
python - <<'PY'
import csv, numpy as np
rows=list(csv.DictReader(open('gpurun_out/prof3/cfg3_kernel_trace.csv')))
ks=[r for r in rows if 'k_apply' in r['Kernel_Name']]
from collections import defaultdict
d=defaultdict(list)
seq=[]
for r in ks:
    dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    seq.append((r['Kernel_Name'][:40],dur))
for i,(n,dur) in enumerate(seq): d[(n,i%3)].append(dur)
for k,v in sorted(d.items()): print(k, len(v), "mean ms %.3f min %.3f"%(np.mean(v), np.min(v)))
PY
find $OUT -type f -size +2M -delete
