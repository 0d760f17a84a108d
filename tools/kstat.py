"""usage: kstat.py run_kernel_stats.csv substring ... -> calls / average us of the kernels whose names contain a substring"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
