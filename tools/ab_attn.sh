#!/bin/bash
# On the GPU box: the attention tile's own launch (per-layer-launch form of the sampler step) for several builds, one line each.
# usage: tools/ab_attn.sh name=path/to/lib.so[,bench args] ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/ab
for cfg in "$@"; do
  name=${cfg%%=*}; lib=${cfg#*=}; extra=""
  case "$lib" in *,*) extra=$(echo "${lib#*,}" | tr ',' ' '); lib=${lib%%,*};; esac
  out=$(DIFFAB_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-other-configs --multi-launch $extra 2>$R/gpurun_out/ab/last.err | tail -1)
  echo "$name: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("attention launch %.4f ms  (step %.4f)" % (r["avg_launch_ms"], d["ms_per_step"]))' 2>/dev/null || tail -3 $R/gpurun_out/ab/last.err)"
done
