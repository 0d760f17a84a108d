#!/bin/bash
# On the GPU box: per-shape durations of the backward GEMM kernels in a short training run (rocprofv3 kernel trace).  usage: tools/tn_shapes.sh TAG
R=$GRAFT_REPO_ROOT; tag=$1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tn_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tn_$tag -o run -- python3 $R/bench.py --train --steps 3 --warmup 3 --no-cpu-baseline > /tmp/tn_$tag.log 2>&1 || { tail -5 /tmp/tn_$tag.log; exit 1; }
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('/tmp/tn_$tag/run_kernel_trace.csv')))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'gemm_tn' in n or 'colsum' in n:
        key=(n.split('(')[0][-26:], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
        agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:8]:
    print(k, len(v), 'avg us %.1f'%(sum(v)/len(v)), 'per step ms %.3f'%(sum(v)/6e3))
PY
