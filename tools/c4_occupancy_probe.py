#!/usr/bin/env python3
"""Config 4's ragged walk against WALKERS PER CU (round 6): the same kernel built for five waves per SIMD (-DGBWT_HIP_WALK_WAVES=5: 96 VGPRs, 21
spilled outside the loops -- ten workgroups per CU where the rings leave room for them) against the product build (four waves, eight
workgroups), each with the default rings (64 slots, 128-byte row pieces) and with half-size ones (32 slots, 64-byte pieces).
One process per library (GBWT_HIP_LIB); usage: c4_occupancy_probe.py SIZE"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4_bench
import gbwt_rs_amd as G
size = sys.argv[1] if len(sys.argv) > 1 else "small"
path = f"/dev/shm/gbwt_c4_occ_{size}.gbz"
if not os.path.exists(path):
    c4_bench.generate(size, path)
generic = np.load(path + ".generic.npy")
gbz = G.GBZ.load(path, flags=G.OPEN_EXTRACT)
walks = np.setdiff1d(np.arange(gbz.paths(), dtype=np.uint64), generic)
ids = 2 * walks
knobs = ["GBWT_HIP_RING_SLOTS", "GBWT_HIP_ROW_PIECE", "GBWT_HIP_HELPER_NAPS"]
for env in ({}, {"GBWT_HIP_RING_SLOTS": "32", "GBWT_HIP_ROW_PIECE": "16"}, {"GBWT_HIP_RING_SLOTS": "32", "GBWT_HIP_ROW_PIECE": "16", "GBWT_HIP_HELPER_NAPS": "2"}, {}):
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
    print(f"{os.path.basename(os.environ.get('GBWT_HIP_LIB', 'libgbwt_hip.so')):22s} {str(env):90s} walk {np.median(wk):7.3f} ms ({int(o.total) / np.median(wk) / 1e6:6.1f} G LF-steps/s)", flush=True)
    w.close()
gbz.close()
