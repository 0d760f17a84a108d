#!/bin/bash
# On the GPU box: the bench's step time for several builds of the library, one line each, same box, back to back.
# usage: tools/ab_variants.sh [-r rounds] name=path/to/libdiffab_hip.so[,bench args] ...   ("base" = the product library)
#        e.g. novpl=diffab-pytorch_amd/lib/libdiffab_hip.so,--attn-variant,16
# Per build: the module-launch form (the headline) and the per-layer-launch form (whose timed kernel is the attention tile's own launch).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=1
if [ "$1" = "-r" ]; then rounds=$2; shift 2; fi
mkdir -p $R/gpurun_out/ab
for ((i = 0; i < rounds; i++)); do
  for cfg in "$@"; do
    name=${cfg%%=*}; lib=${cfg#*=}; extra=""
    [ "$cfg" = base ] && lib=$R/diffab-pytorch_amd/lib/libdiffab_hip.so
    case "$lib" in *,*) extra=$(echo "${lib#*,}" | tr ',' ' '); lib=${lib%%,*};; esac
    for form in "" "--multi-launch"; do
      out=$(DIFFAB_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-other-configs $form $extra 2>$R/gpurun_out/ab/last.err | tail -1)
      echo "$name ${form:---module}: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms/step %.4f  timed kernel %.4f ms x %d (%s)" % (d["ms_per_step"], r["avg_launch_ms"], r["launches"], r["kernel"][:28]))' 2>/dev/null || tail -3 $R/gpurun_out/ab/last.err)"
    done
  done
done
