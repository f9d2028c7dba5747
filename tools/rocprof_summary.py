#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (or kernel-trace csv) into the plain-text summary kept under profiles/."""
import csv
import glob
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    return [(n, int(c), float(t), float(a), float(p)) for n, c, t, a, p in rows]


def from_csv(path):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        agg[r["Kernel_Name"]][0] += 1
        agg[r["Kernel_Name"]][1] += d
    total = sum(v[1] for v in agg.values()) or 1.0
    return sorted(((n, c, t, t / c, 100 * t / total) for n, (c, t) in agg.items()), key=lambda x: -x[2])


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'calls':>7} {'total_us':>14} {'avg_us':>14} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        scale = 1e-3 if not path.endswith(".db") else 1.0   # csv timestamps are ns, the db view is us
        print(f"{calls:7d} {total * scale:14.1f} {avg * scale:14.1f} {pct:7.2f}  {short}")


if __name__ == "__main__":
    main()
