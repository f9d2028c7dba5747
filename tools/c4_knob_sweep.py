#!/usr/bin/env python3
"""Config 4 (size from argv, default full): the walk kernel under workspace knobs (read when a workspace is created): one open, one workspace per setting."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4_bench
import gbwt_rs_amd as G
size = sys.argv[1] if len(sys.argv) > 1 else "full"
path = "/dev/shm/gbwt_c4_knobs.gbz"
g = c4_bench.generate(size, path)
generic = np.load(path + ".generic.npy")
walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
ids = 2 * walks
gbz = G.GBZ.load(path, flags=G.OPEN_EXTRACT)
knobs = ["GBWT_HIP_UNIFORM_LOOP", "GBWT_HIP_CATCH_UP", "GBWT_HIP_HELPER_NAPS", "GBWT_HIP_RING_SLOTS", "GBWT_HIP_XCD_MAP", "GBWT_HIP_ALL4", "GBWT_HIP_VMM"]
settings = [{}, {"GBWT_HIP_UNIFORM_LOOP": "0"}, {"GBWT_HIP_CATCH_UP": "0"}, {"GBWT_HIP_UNIFORM_LOOP": "0", "GBWT_HIP_CATCH_UP": "0"}, {"GBWT_HIP_HELPER_NAPS": "2"}, {"GBWT_HIP_HELPER_NAPS": "8"},
            {"GBWT_HIP_XCD_MAP": "0"}, {"GBWT_HIP_RING_SLOTS": "128"}, {"GBWT_HIP_VMM": "0"}, {}]
for env in settings:
    for k in knobs:
        os.environ.pop(k, None)
    os.environ.update(env)
    w = gbz.another_workspace()
    for _ in range(3):
        w.extract_device(ids)
    wk = []
    for _ in range(5):
        o = w.extract_device(ids)
        wk.append(w.last_kernel_ms()[0])
    print(f"{str(env):70s} walk {np.median(wk):7.3f} ms ({int(o.total) / np.median(wk) / 1e6:6.1f} G LF-steps/s)", flush=True)
    w.close()
gbz.close()
c4_bench.cleanup(path)
