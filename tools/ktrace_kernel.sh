#!/bin/bash
# rocprofv3 kernel trace of a short bench run: every launch of the kernels whose name contains $1 (duration in µs, in launch order,
# grid size).  usage: tools/ktrace_kernel.sh <name part> [bench args]
pat=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-other-configs "$@" > /tmp/kt_bench.json 2>/dev/null
python3 - "$pat" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open("/tmp/kt/run_kernel_trace.csv")) if sys.argv[1] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-12:]:
    print(f'{r["Kernel_Name"][:60]:60s} grid {r["Grid_Size_X"]:>7s} x {r["Grid_Size_Y"]:>2s}  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us')
PY
