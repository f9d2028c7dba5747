#!/usr/bin/env python3
"""Config 4 at its stated size: the walk kernel under different sample intervals / checkpoint gaps (read at open), one generation, one open per setting."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4_bench
import gbwt_rs_amd as G
size = sys.argv[1] if len(sys.argv) > 1 else "full"
path = "/dev/shm/gbwt_c4_sweep.gbz"
g = c4_bench.generate(size, path)
generic = np.load(path + ".generic.npy")
walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
ids = 2 * walks
settings = [{}] + [{"GBWT_HIP_SAMPLE_INTERVAL": str(i)} for i in (256, 512, 1024, 4096)] + \
           [{"GBWT_HIP_CHECKPOINT_GAP": str(gap)} for gap in (64, 256, 512)] + [{"GBWT_HIP_SAMPLE_INTERVAL": "1024", "GBWT_HIP_CHECKPOINT_GAP": "256"}, {"GBWT_HIP_CHAINS": "0"}]
for env in settings:
    for k in ("GBWT_HIP_SAMPLE_INTERVAL", "GBWT_HIP_CHECKPOINT_GAP", "GBWT_HIP_CHAINS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    gbz = G.GBZ.load(path, flags=G.OPEN_EXTRACT)
    open_s = time.perf_counter() - t0
    for _ in range(2):
        gbz.extract_device(ids)
    wk = []
    for _ in range(4):
        o = gbz.extract_device(ids)
        wk.append(gbz.last_kernel_ms()[0])
    ot = gbz.open_times()
    print(f"{str(env):70s} walk {np.median(wk):7.3f} ms ({int(o.total) / np.median(wk) / 1e6:6.1f} G LF-steps/s)  open {open_s:5.2f} s  samples {int(ot['samples'])}  device {gbz.memory_usage()['index_device_bytes'] / 1e9:.1f} GB", flush=True)
    gbz.close()
c4_bench.cleanup(path)
