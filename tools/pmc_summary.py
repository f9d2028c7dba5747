#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc output directory (counter_collection.csv files)."""
import csv
import glob
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
for k, cs in agg.items():
    n = len(calls[k])
    print(f"{k}  dispatches={n}")
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {v / n:18.1f} per dispatch")
