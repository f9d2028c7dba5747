#!/usr/bin/env python3
"""Which mutations of tests/test_capi_cpu.py: mutated_large_file make an open slow (seconds per mutation, the slow ones listed)."""
import os, sys, time, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gbwt_rs_amd as G
from test_capi_cpu import mutated_large_file
per_region = int(sys.argv[1]) if len(sys.argv) > 1 else 28
tmp = pathlib.Path(tempfile.mkdtemp())
t_all = time.perf_counter()
rows = []
for w, value, path in mutated_large_file(tmp, per_region=per_region):
    t0 = time.perf_counter()
    try:
        dev = G.GBZ.load(path)
        t1 = time.perf_counter()
        dev.sequences_csr(np.arange(0, min(dev.sequences(), 64), dtype=np.uint64))
        t2 = time.perf_counter()
        dev.close()
        rows.append((time.perf_counter() - t0, w, hex(value), "opened", t1 - t0, t2 - t1))
    except G.GbwtHipError as e:
        rows.append((time.perf_counter() - t0, w, hex(value), str(e)[:50], 0, 0))
print("mutations", len(rows), "total", round(time.perf_counter() - t_all, 1), "s")
for r in sorted(rows, reverse=True)[:25]:
    print(r)
