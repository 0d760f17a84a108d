#!/bin/bash
# rocprofv3 kernel trace of a short bench run: every launch of the kernels whose name contains $1 (duration in µs, in launch order,
# grid size).  usage: tools/ktrace_kernel.sh <name part> [bench args]   (single-GPU bench arguments only: under rocprofv3 bench.py must
# not start ranks of its own - the profiler has initialised the GPU and the box refuses the fork + exec)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
pat=$1; shift
for a in "$@"; do
  case "$a" in --gpus|--gpus=*) echo "ktrace_kernel.sh: --gpus is not supported under rocprofv3" >&2; exit 2;; esac
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
if ! rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o run -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-other-configs "$@" > /tmp/kt_bench.json 2>/tmp/kt_bench.err; then
  echo "ktrace_kernel.sh: the profiled bench run failed:" >&2; tail -5 /tmp/kt_bench.err >&2; exit 1
fi
python3 - "$pat" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open("/tmp/kt/run_kernel_trace.csv")) if sys.argv[1] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-12:]:
    print(f'{r["Kernel_Name"][:60]:60s} grid {r["Grid_Size_X"]:>7s} x {r["Grid_Size_Y"]:>2s}  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us')
PY
