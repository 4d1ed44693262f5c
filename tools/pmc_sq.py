"""Sums SQ counters per kernel name from a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run directory.
python tools/pmc_sq.py <dir>   (reads */*counter_collection.csv)"""
import csv, glob, sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k, c in rows.items():
    n = len(calls[k])
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k}  x{n}")
    for name, v in sorted(c.items()):
        print(f"    {name:28s} {v / n:14.0f}  {100 * v / wc:6.1f}% of WAVE_CYCLES")
