#!/usr/bin/env python3
"""Whole-file write of config 4's stand-in (4.5 GB to /dev/shm) under GBWT_HIP_GFA_WRITERS = 3, 8, 16, ...: seconds and GB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import c4_bench
keep = {}
c4_bench.run(size="small", passes=1, keep=keep)
gbz = keep["gbz"]
out = "/dev/shm/gbwt_writers_sweep.gfa"
for writers in [int(x) for x in (sys.argv[1:] or ["3", "8", "16", "32", "64"])]:
    os.environ["GBWT_HIP_GFA_WRITERS"] = str(writers)
    for rep in range(2):
        if os.path.exists(out):
            os.remove(out)
        t0 = time.perf_counter()
        gbz.write_gfa(out)
        dt = time.perf_counter() - t0
        print(f"writers {writers:2d}: {os.path.getsize(out)} bytes in {dt:.3f} s = {os.path.getsize(out) / dt / 1e9:.2f} GB/s", flush=True)
os.remove(out)
c4_bench.cleanup(keep["path"])
