#!/usr/bin/env python3
"""Config 4 (small stand-in, or the size in argv): five extractions of all forward sequences of the walks -- the ragged batch whose waves are
mixed (gather form of the two-step loop) -- for a counter pass over k_walk_direct (tools/walk_pmc.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4_bench
import gbwt_rs_amd as G
size = sys.argv[1] if len(sys.argv) > 1 else "small"
path = "/dev/shm/gbwt_c4_walk_pmc.gbz"
g = c4_bench.generate(size, path)
generic = np.load(path + ".generic.npy")
walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
gbz = G.GBZ.load(path, flags=G.OPEN_EXTRACT)
for _ in range(5):
    out = gbz.extract_device(2 * walks)
print("LF-steps per extraction", int(out.total), "kernel ms", gbz.last_kernel_ms()[0], flush=True)
gbz.close()
c4_bench.cleanup(path)
