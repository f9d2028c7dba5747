#!/usr/bin/env python3
"""Config 4 (small stand-in): after the request that fills the line cache, three W-line requests of all walks under GBWT_HIP_FORMAT_TOKENS=0 and
three under =1 -- the two instantiations of k_format_chunks side by side in one counter pass (tools/format_pmc.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4_bench
import gbwt_rs_amd as G
path = "/dev/shm/gbwt_c4_pmc.gbz"
g = c4_bench.generate("small", path)
generic = np.load(path + ".generic.npy")
walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
gbz = G.GBZ.load(path, flags=G.OPEN_GFA)
gbz.path_lines_device(walks, 1)
for mode in ("0", "1"):
    os.environ["GBWT_HIP_FORMAT_TOKENS"] = mode
    for _ in range(3):
        gbz.path_lines_device(walks[:1], 1)
        out = gbz.path_lines_device(walks, 1)
print("text bytes per request", int(out.total), "positions", int((gbz.len() - gbz.sequences()) // 2), flush=True)
gbz.close()
c4_bench.cleanup(path)
