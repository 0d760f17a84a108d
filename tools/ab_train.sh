#!/bin/bash
# On the GPU box: the training step (bench.py --train, BASELINE config 4's per-GPU share) for several library builds / variants, same box.
# usage: tools/ab_train.sh [-r rounds] name=path/to/libdiffab_hip.so[,bench args] ...   ("base" = the product library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=1
if [ "$1" = "-r" ]; then rounds=$2; shift 2; fi
mkdir -p $R/gpurun_out/ab
for ((i = 0; i < rounds; i++)); do
  for cfg in "$@"; do
    name=${cfg%%=*}; lib=${cfg#*=}; extra=""
    [ "$cfg" = base ] && lib=$R/diffab-pytorch_amd/lib/libdiffab_hip.so
    case "$lib" in *,*) extra=$(echo "${lib#*,}" | tr ',' ' '); lib=${lib%%,*};; esac
    out=$(DIFFAB_HIP_LIB=$lib timeout -k 10 200 python3 $R/bench.py --train --steps 10 --warmup 3 --no-cpu-baseline $extra 2>$R/gpurun_out/ab/last.err | tail -1)
    echo "$name train: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms/step %.4f with d pair_ctx, %.4f contexts constant" % (d["ms_per_step"], d["contexts_constant"]["ms_per_step"]))' 2>/dev/null || tail -3 $R/gpurun_out/ab/last.err)"
  done
done
