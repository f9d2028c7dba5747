#!/usr/bin/env python3
"""HBM traffic of one or more kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass)
-> the JSON kept under profiles/ and quoted by bench.py as roofline.traffic.

usage: hbm_traffic.py FETCH_DIR WRITE_DIR KERNEL[,KERNEL...] "workload text" ALGORITHMIC_BYTES [--key KEY] [--pick last|max] [--fetch-factor F] [bench.py arguments]
           > profiles/rNN_x_hbm_traffic.json
--key: the workload key bench.py looks the profile up by (default: the headline's "sites=... haplotypes=... model=... seed=...";
the other configs: "secondary", "high_degree", "search", "config4").  --pick: which dispatch of a kernel counts -- the last one
(default: the timed launch behind the warm-ups) or the largest (commands whose launches differ in size).  With several kernels the
top-level figures are their SUM (the workload's traffic); every kernel's own are under "kernels".
The JSON carries bench.py's fingerprint of the kernel sources + GBWT_HIP_* knobs: bench.py quotes the traffic only for a run of the
same build, knobs and workload.
Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of
wide coalesced 16 B/lane loads -> doubled (--fetch-factor 2, the default); WRITE_SIZE is uncalibrated and taken as is.  Both are in KiB.
Round 5 calibrated the counter on scattered reads (tools/microbench_fetch.hip, profiles/r05_fetch_calibration.txt): it counts 64 bytes per
REQUEST, so a lane's scattered 64-byte line or 16-byte entry is counted as the 64 bytes that move -- the search kernels' profile is taken
with --fetch-factor 1."""
import csv
import glob
import json
import os
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


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

fetch_dir, write_dir, kernels, workload, algorithmic = sys.argv[1:6]
rest, key, pick, factor = sys.argv[6:], None, "last", 2.0
while rest and rest[0] in ("--key", "--pick", "--fetch-factor"):
    if rest[0] == "--key":
        key = rest[1]
    elif rest[0] == "--fetch-factor":
        factor = float(rest[1])
    else:
        pick = rest[1]
    rest = rest[2:]
sys.argv = [sys.argv[0]] + rest
bench_args = bench.parse_args()
choose = (lambda v: v[-1]) if pick == "last" else max
per_kernel, f_sum, w_sum = {}, 0.0, 0.0
for kernel in kernels.split(","):
    fetch, write = dispatches(fetch_dir, "FETCH_SIZE", kernel), dispatches(write_dir, "WRITE_SIZE", kernel)
    f_raw, w_raw = choose(fetch) * 1024, choose(write) * 1024
    per_kernel[kernel] = {"dispatches": {"FETCH_SIZE_KiB": fetch[-8:], "WRITE_SIZE_KiB": write[-8:]}, "fetch_bytes_raw": f_raw, "fetch_bytes_corrected": factor * f_raw,
                          "write_bytes": w_raw, "traffic_bytes_per_launch": factor * f_raw + w_raw}
    f_sum += f_raw
    w_sum += w_raw
first = per_kernel[kernels.split(",")[0]]
print(json.dumps({
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes of the same command)",
    "workload": workload,
    "kernel": kernels,
    "unit_note": "FETCH_SIZE / WRITE_SIZE are in KiB",
    "dispatches": first["dispatches"],
    "fetch_bytes_raw": f_sum,
    "fetch_bytes_corrected": factor * f_sum,
    "fetch_factor": factor,
    "correction": ("MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced 16 B/lane loads -> doubled; WRITE_SIZE uncalibrated, taken as is"
                   if factor == 2.0 else
                   "profiles/r05_fetch_calibration.txt: FETCH_SIZE counts one 64-byte unit per REQUEST -- exact for the scattered 64-byte lines (descriptors) and "
                   "16-byte entries (rank blocks: a whole 64-byte line moves) a lane of the search kernels reads, half only for 128-byte requests: taken as is"),
    "write_bytes": w_sum,
    "traffic_bytes_per_launch": factor * f_sum + w_sum,
    "algorithmic_bytes_per_launch": float(algorithmic),
    "kernels": per_kernel,
    "pick": pick,
    "source_fingerprint": bench.source_fingerprint(),
    "workload_key": key or f"sites={bench_args.sites} haplotypes={bench_args.haplotypes} model={bench_args.model} seed={bench_args.seed}",
}, indent=1))
