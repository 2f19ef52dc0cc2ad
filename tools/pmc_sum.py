"""Sums rocprofv3 --pmc counter_collection.csv per kernel and counter:  python tools/pmc_sum.py <csv> [kernel substring]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(float); n = defaultdict(int)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r.get("Kernel_Name", "")
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        k = k.split("(")[0][-60:]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(acc.items()):
    print(f"{k:62s} {c:28s} launches {n[(k, c)]:4d}  per-launch {v / n[(k, c)]:.4g}")
