#!/usr/bin/env python3
"""Counters of a rocprofv3 --pmc output directory SUMMED over all dispatches of each kernel whose name contains one of the given words
(tiny dispatches next to the big ones then do not halve an average)."""
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("gbwt_hip::", "").split("(")[0]
        if not any(w in k for w in sys.argv[2:]):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
for k, cs in sorted(agg.items()):
    print(f"{k}  dispatches={len(calls[k])}")
    for c, v in sorted(cs.items()):
        print(f"    {c:30s} {v:18.0f}")
