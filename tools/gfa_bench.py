#!/usr/bin/env python3
"""W-line text of all paths of a synthetic GBZ on one GPU (gbwt_hip_path_lines: extraction + formatting on the device,
text copied to the host): bytes, wall time of the size query (walk + token widths) and of the full call."""
import argparse
import hashlib
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
args = ap.parse_args()

s = S.Synth.chain(args.sites, args.haplotypes, alleles=2, model=S.MOSAIC, seed=42)
path = os.path.join(tempfile.mkdtemp(prefix="gfa_bench_"), "bench.gbz")
s.save(path, as_gbz=True)
gbz = G.GBZ.load(path)
ids = np.arange(gbz.paths(), dtype=np.uint64)
nodes = (gbz.len() - gbz.sequences()) // 2
for rep in range(3):
    t0 = time.perf_counter()
    text = gbz.path_lines(ids, 1)
    dt = time.perf_counter() - t0
    print(f"W-lines of {len(ids)} paths, {nodes} nodes: {len(text)} bytes in {dt * 1e3:.1f} ms = {len(text) / dt / 1e9:.2f} GB/s of text, "
          f"{nodes / dt / 1e9:.2f} G nodes/s   sha256 {hashlib.sha256(text).hexdigest()[:16]}", flush=True)
# into a numpy array (no zero-filled ctypes buffer, no bytes copy): what the C call itself costs -- format once, then the copy to the host
# over gbwt_hip's pinned staging threads; a fresh destination pays its first-touch page faults, a reused one does not
buf = None
for rep in range(4):
    gbz.path_lines_device(ids[:1], 1)                    # (another request in between: the lines are formatted again)
    t0 = time.perf_counter()
    arr = gbz.path_lines_array(ids, 1, out=buf)
    dt = time.perf_counter() - t0
    walk_ms, fmt_ms = gbz.last_lines_ms()
    print(f"  into {'a fresh' if buf is None else 'the same'} numpy array: {arr.size} bytes in {dt * 1e3:.1f} ms = {arr.size / dt / 1e9:.2f} GB/s of text "
          f"(walk {walk_ms:.2f} ms + format {fmt_ms:.2f} ms on the device; the rest is the copy: {arr.size / max(1e-9, dt - (walk_ms + fmt_ms) * 1e-3) / 1e9:.1f} GB/s)", flush=True)
    if rep >= 1:
        buf = arr.base if arr.base is not None else arr
assert hashlib.sha256(arr.tobytes()).hexdigest() == hashlib.sha256(text).hexdigest()
for rep in range(3):
    t0 = time.perf_counter()
    lines = gbz.path_lines_device(ids[::-1].copy(), 1)   # another request: nothing of the calls above is reused
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    lines = gbz.path_lines_device(ids, 1)
    dt2 = time.perf_counter() - t0
    print(f"the same lines left in HBM (gbwt_hip_path_lines_device): {lines.total} bytes in {dt2 * 1e3:.1f} ms = {lines.total / dt2 / 1e9:.1f} GB/s of text, "
          f"{nodes / dt2 / 1e9:.1f} G nodes/s (reversed order: {dt * 1e3:.1f} ms)", flush=True)
os.remove(path)
