#!/bin/bash
# On the GPU box: grid / block sizes and mean durations of every kernel of a short training run (rocprofv3 kernel trace).  usage: tools/train_grids.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tg
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o run -- python3 $R/bench.py --train --steps 3 --warmup 3 --no-cpu-baseline > /tmp/tg.log 2>&1 || { tail -5 /tmp/tg.log; exit 1; }
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('/tmp/tg/run_kernel_trace.csv')))
agg=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].split('(')[0].replace('void diffab::','').replace('diffab::','')
    if 'at::native' in n or 'rocclr' in n: continue
    wg=int(r['Workgroup_Size_X'])*int(r['Workgroup_Size_Y'])*int(r['Workgroup_Size_Z'])
    key=(n[:44], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), int(r['Grid_Size_Y'])//int(r['Workgroup_Size_Y']), int(r['Grid_Size_Z'])//int(r['Workgroup_Size_Z']), wg, r.get('LDS_Block_Size','?'))
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:26]:
    print(f"{k[0]:44s} grid {k[1]:5d} x{k[2]:4d} x{k[3]:3d} wg {k[4]:4d} lds {k[5]:>6s}  n {len(v):3d} avg us {sum(v)/len(v):7.1f}  ms/step {sum(v)/6e3:6.3f}")
PY
