#!/usr/bin/env python3
"""FETCH_SIZE per dispatch of tools/microbench_fetch (rocprofv3 --pmc FETCH_SIZE output directory) next to the bytes the lanes asked for."""
import csv, glob, sys
lanes = 1 << 25
asked = {"k_scattered<1>#0": lanes * 16, "k_scattered<1>#1": lanes * 16, "k_scattered<4>": lanes * 64, "k_scattered<8>": lanes * 128, "k_streaming": lanes * 16}
rows = {}
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
seen = {}
for d in sorted(rows):
    name, kib = rows[d]
    if "k_streaming" not in name and "k_scattered" not in name:
        continue
    short = "k_streaming" if "k_streaming" in name else ("k_scattered<" + name.split("k_scattered<")[1].split(">")[0] + ">")
    if short == "k_scattered<1>":
        short += "#" + str(seen.get("one", 0) % 2)
        seen["one"] = seen.get("one", 0) + 1
    want = asked.get(short)
    print(f"dispatch {d:3d} {short:18s} FETCH_SIZE {kib * 1024 / 1e9:8.3f} GB   asked {want / 1e9 if want else 0:8.3f} GB   counter / asked = {kib * 1024 / want if want else 0:.3f}")
