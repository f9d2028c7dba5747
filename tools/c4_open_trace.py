#!/usr/bin/env python3
"""GBWT_HIP_TRACE_OPEN=1 over config 4's GBZ (size from argv): where the open goes, flags ALL and GFA."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4_bench
import gbwt_rs_amd as G
size = sys.argv[1] if len(sys.argv) > 1 else "full"
path = "/dev/shm/gbwt_c4_open.gbz"
g = c4_bench.generate(size, path)
del g
os.environ["GBWT_HIP_TRACE_OPEN"] = "1"
for flags, name in ((G.OPEN_ALL, "ALL"), (G.OPEN_GFA, "GFA"), (G.OPEN_GFA, "GFA again")):
    print(f"==== open with {name}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    gbz = G.GBZ.load(path, flags=flags)
    dt = time.perf_counter() - t0
    ot = gbz.open_times()
    print(f"{name}: {dt * 1e3:.0f} ms wall; parse {ot['parse_ms']:.0f}, upload {ot['upload_ms']:.0f}, samples {ot['sample_ms']:.0f}; device {gbz.memory_usage()['index_device_bytes'] / 1e9:.1f} GB", flush=True)
    gbz.close()
c4_bench.cleanup(path)
