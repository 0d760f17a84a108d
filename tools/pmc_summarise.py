"""Average rocprofv3 --pmc counters per kernel launch.  usage: pmc_summarise.py OUT.json DIR [DIR ...]
Each DIR holds one pass's *_counter_collection.csv; kernels are keyed by name without the argument list."""
import csv
import glob
import json
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))  # kernel -> counter -> [sum over launches, launches]
for d in dirs:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = defaultdict(float)  # (dispatch id, kernel, counter) -> value summed over XCDs/instances
        for r in csv.DictReader(open(path)):
            per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per_dispatch.items():
            acc[k][c][0] += v
            acc[k][c][1] += 1
res = {k: {c: s / n for c, (s, n) in cs.items()} for k, cs in acc.items() if "diffab" in k}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, cs in res.items():
    if "attn" in k or "proj_frames" in k or "rowgemm" in k or "gemm" in k or "module" in k:
        print(k, {c: round(v, 1) for c, v in cs.items()})
