#!/bin/bash
# kernel stats of the queue-driven launch for several builds (tools/ablate.sh names) - usage: queue_prof2.sh name...
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  export DIFFAB_HIP_LIB=$GRAFT_REPO_ROOT/diffab-pytorch_amd/build_abl/$name/libdiffab_hip.so DIFFAB_ATTN_QUEUE=1 DIFFAB_ATTN_QUEUE_STAGGER=${QS:-0}
  rm -rf /tmp/qp
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qp -o run -- python3 $GRAFT_REPO_ROOT/tools/queue_check.py ${QB:-256} child /tmp/qp_out.pt > /tmp/qp.log 2>&1
  echo "== $name: $(grep layer /tmp/qp.log)"
  python3 - <<'P'
import csv
for r in csv.DictReader(open("/tmp/qp/run_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("ipa_attn", "rowgemm128_b6")):
        print("   ", r["Name"].split("(")[0][:60].ljust(60), r["Calls"], "avg %.1f us  min %.1f us" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
P
done
