#!/bin/bash
# On the GPU box: kernel stats of one IPA layer, default launches vs the queue-driven launch (EXPERIMENTAL build) at several staggers.
cd /tmp && export TMPDIR=/tmp
export DIFFAB_HIP_LIB=$GRAFT_REPO_ROOT/experiments/build/libdiffab_hip.so
for cfg in "0 0" "1 0" "1 250" "1 500" "1 1000"; do
  set -- $cfg
  export DIFFAB_ATTN_QUEUE=$1 DIFFAB_ATTN_QUEUE_STAGGER=$2
  rm -rf /tmp/qp
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qp -o run -- python3 $GRAFT_REPO_ROOT/tools/queue_check.py ${QB:-256} child /tmp/qp_out.pt > /tmp/qp.log 2>&1
  echo "== queue=$1 stagger=$2: $(grep layer /tmp/qp.log)"
  python3 - <<'P'
import csv
for r in csv.DictReader(open("/tmp/qp/run_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("ipa_attn", "rowgemm128_b6", "proj_frames")):
        print("   ", r["Name"].split("(")[0][:60].ljust(60), r["Calls"], "avg %.1f us  min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
P
done
