#!/usr/bin/env python3
"""W-lines of all paths of a synthetic GBZ left in HBM (gbwt_hip_path_lines_device), for kernel profiles:
rocprofv3 --kernel-trace --stats -- python3 tools/gfa_device_bench.py"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=20000)
ap.add_argument("--haplotypes", type=int, default=5000)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
s = S.Synth.chain(args.sites, args.haplotypes, alleles=2, model=S.MOSAIC, seed=42)
path = os.path.join(tempfile.mkdtemp(prefix="gfa_bench_"), "bench.gbz")
s.save(path, as_gbz=True)
gbz = G.GBZ.load(path)
ids = np.arange(gbz.paths(), dtype=np.uint64)
for rep in range(args.reps):
    t0 = time.perf_counter()
    lines = gbz.path_lines_device(ids[::-1].copy() if rep % 2 else ids, 1)
    dt = time.perf_counter() - t0
    print(f"{lines.total} bytes of W-lines in {dt * 1e3:.2f} ms = {lines.total / dt / 1e9:.1f} GB/s", flush=True)
os.remove(path)
