#!/bin/bash
# On the GPU box: time every kernel of one IPA layer for each ablation build (rocprofv3 kernel stats).  tools/abl_run.sh name...
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  export DIFFAB_HIP_LIB=$GRAFT_REPO_ROOT/diffab-pytorch_amd/build_abl/$name/libdiffab_hip.so
  rm -rf /tmp/abl_$name
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$name -o run -- python3 $GRAFT_REPO_ROOT/tools/${ABL_TOOL:-attn_phase_profile.py} ${ABL_B:-256} ${ABL_K:-128} > /tmp/abl_$name.log 2>&1
  echo "== $name"; grep -E "phase1|phase2 prol|phase2 \(|phase3|lifetime" /tmp/abl_$name.log
  python3 - "$name" <<'P'
import csv, sys
for r in csv.DictReader(open(f"/tmp/abl_{sys.argv[1]}/run_kernel_stats.csv")):
    if "diffab" in r["Name"] and "igso3" not in r["Name"]:
        print("  ", r["Name"].split("(")[0][:50].ljust(50), r["Calls"], "avg %.1f us  min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
P
done
