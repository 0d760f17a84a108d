#!/bin/bash
# Two-queue wrong-result investigation: tools/step_race_check.py with each heads_finish variant of the EXPERIMENTAL build.
R=$GRAFT_REPO_ROOT
export DIFFAB_HIP_LIB=$R/experiments/build/libdiffab_hip.so
O=$R/gpurun_out/hf; mkdir -p $O
for v in ${VARS:-0 5 1 2 3 4}; do
  echo "== DIFFAB_HF_VARIANT=$v" | tee -a $O/hf.log
  DIFFAB_HF_VARIANT=$v timeout -k 10 200 python3 $R/tools/step_race_check.py 128 ${REPS:-300} 2>&1 | grep -v Warning | tee -a $O/hf.log | tail -6
done
[ -n "$SKIP64" ] && exit 0; echo "== block 64" | tee -a $O/hf.log
DIFFAB_HF_BLOCK=64 timeout -k 10 200 python3 $R/tools/step_race_check.py 128 ${REPS:-300} 2>&1 | tee -a $O/hf.log | tail -6
