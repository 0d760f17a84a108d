#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of a short bench run; prints the top kernels.  usage: tools/kstats.sh TAG [bench args...]
R=$GRAFT_REPO_ROOT; tag=$1; shift
O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o run -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs "$@" > $O/ks_$tag.log 2>&1 || { tail -5 $O/ks_$tag.log; exit 1; }
cp /tmp/ks_$tag/run_kernel_stats.csv $O/kernel_stats_$tag.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_$tag.csv")))
steps=23
for r in rows[:14]:
    n=r['Name'].split('(')[0].replace('void diffab::','').replace('diffab::','')[:58]
    print(f"{n:58s} calls/step {int(r['Calls'])/steps:5.1f}  avg us {float(r['AverageNs'])/1e3:8.1f}  ms/step {float(r['TotalDurationNs'])/steps/1e6:6.3f}")
PY
