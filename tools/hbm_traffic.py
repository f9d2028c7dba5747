#!/usr/bin/env python3
"""HBM traffic of the walk kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass)
-> the JSON kept under profiles/ and quoted by bench.py as roofline.traffic.

usage: hbm_traffic.py FETCH_DIR WRITE_DIR KERNEL "workload text" OUTPUT_BYTES [bench.py arguments] > profiles/rNN_x_hbm_traffic.json
The JSON carries bench.py's fingerprint of the kernel sources + GBWT_HIP_* knobs and the workload key: bench.py quotes
the traffic only for a run of the same build, knobs and workload.
Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of
16 B/lane loads -> doubled; WRITE_SIZE is uncalibrated and taken as is.  Both are in KiB."""
import csv
import glob
import json
import sys


def dispatches(directory, counter, kernel):
    vals = []
    for path in sorted(glob.glob(directory + "/**/*counter_collection.csv", recursive=True)):
        per = {}
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        vals += [per[k] for k in sorted(per)]
    return vals


import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

fetch_dir, write_dir, kernel, workload, algorithmic = sys.argv[1:6]
sys.argv = [sys.argv[0]] + sys.argv[6:]
bench_args = bench.parse_args()
fetch, write = dispatches(fetch_dir, "FETCH_SIZE", kernel), dispatches(write_dir, "WRITE_SIZE", kernel)
f_raw, w_raw = fetch[-1] * 1024, write[-1] * 1024      # the last launch = the timed one (the first is the warm-up)
print(json.dumps({
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline",
    "workload": workload,
    "kernel": kernel,
    "unit_note": "FETCH_SIZE / WRITE_SIZE are in KiB",
    "dispatches": {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write},
    "fetch_bytes_raw": f_raw,
    "fetch_bytes_corrected": 2 * f_raw,
    "correction": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of 16 B/lane loads -> doubled; WRITE_SIZE uncalibrated, taken as is",
    "write_bytes": w_raw,
    "traffic_bytes_per_launch": 2 * f_raw + w_raw,
    "algorithmic_bytes_per_launch": float(algorithmic),
    "source_fingerprint": bench.source_fingerprint(),
    "workload_key": f"sites={bench_args.sites} haplotypes={bench_args.haplotypes} model={bench_args.model} seed={bench_args.seed}",
}, indent=1))
