#!/bin/bash
# rocprofv3 kernel stats of a short bench run: the top kernels by total time (µs per launch).  usage: tools/kstats_top.sh [bench args]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs "$@" > /tmp/ks_bench.json 2>/dev/null
python3 - <<'PY'
import csv, json
rows = list(csv.DictReader(open("/tmp/ks/run_kernel_stats.csv")))
import os
for r in rows[:int(os.environ.get("TOP", "7"))]:
    print(f'{r["Name"][:70].replace("void diffab::", ""):72s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:9.1f} us')
try:
    d = json.loads([l for l in open("/tmp/ks_bench.json") if l.startswith("{")][-1])
    print("ms_per_step (under the profiler)", round(d["ms_per_step"], 3))
except Exception as ex:
    print("no bench line", ex)
PY
