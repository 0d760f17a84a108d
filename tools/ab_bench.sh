#!/bin/bash
# On the GPU box: short bench runs under different settings, one line each.
# usage: tools/ab_bench.sh "VAR=a" "DIFFAB_HIP_LIB=path/to/variant.so" ...   (each argument: env assignments for one run, or "base")
R=${GRAFT_REPO_ROOT:-.}
mkdir -p $R/gpurun_out/r2
for cfg in "$@"; do
  envs=$cfg; [ "$cfg" = base ] && envs="DIFFAB_AB=base"
  out=$(env $envs timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>$R/gpurun_out/r2/ab_last.err | tail -1)
  echo "$cfg :: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.4f  attn %.4f ms  frac %.4f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))' 2>/dev/null || tail -3 $R/gpurun_out/r2/ab_last.err)"
done
