#!/usr/bin/env python3
"""Config 4's ragged walk against the look-ahead of mixed waves (GBWT_HIP_GATHER_REACH, WalkArgs::gather_reach; round 6) and the helper's naps,
one open, one workspace per setting; also the insertion chain (every wave mixed, 3 300 rows per record) where it must not be the default.
usage: c4_reach_probe.py SIZE"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4_bench
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
size = sys.argv[1] if len(sys.argv) > 1 else "small"

def sweep(label, dev, ids, settings, check=None):
    knobs = ["GBWT_HIP_GATHER_REACH", "GBWT_HIP_HELPER_NAPS"]
    for env in settings:
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update(env)
        w = dev.another_workspace()
        for _ in range(3):
            w.extract_device(ids)
        wk = []
        for _ in range(5):
            o = w.extract_device(ids)
            wk.append(w.last_kernel_ms()[0])
        ok = "" if check is None else (" sums ok" if np.array_equal(w.path_sums(len(ids)), check) else " SUMS DIFFER")
        print(f"{label:10s} {str(env):70s} walk {np.median(wk):7.3f} ms ({int(o.total) / np.median(wk) / 1e6:6.1f} G LF-steps/s){ok}", flush=True)
        w.close()
    for k in knobs:
        os.environ.pop(k, None)

path = f"/dev/shm/gbwt_c4_reach_{size}.gbz"
g = c4_bench.generate(size, path)
generic = np.load(path + ".generic.npy")
gbz = G.GBZ.load(path, flags=G.OPEN_EXTRACT)
walks = np.setdiff1d(np.arange(gbz.paths(), dtype=np.uint64), generic)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as pool:
    truth = np.array(list(pool.map(g.path_checksum, [int(p) for p in walks])), dtype=np.uint64)
reach = [{"GBWT_HIP_GATHER_REACH": "0"}, {"GBWT_HIP_GATHER_REACH": "1"}, {"GBWT_HIP_GATHER_REACH": "4"}, {"GBWT_HIP_GATHER_REACH": "8"}, {"GBWT_HIP_GATHER_REACH": "16"},
         {"GBWT_HIP_GATHER_REACH": "8", "GBWT_HIP_HELPER_NAPS": "2"}, {"GBWT_HIP_GATHER_REACH": "8", "GBWT_HIP_HELPER_NAPS": "1"}, {"GBWT_HIP_GATHER_REACH": "4", "GBWT_HIP_HELPER_NAPS": "1"},
         {"GBWT_HIP_GATHER_REACH": "1", "GBWT_HIP_HELPER_NAPS": "1"}, {"GBWT_HIP_GATHER_REACH": "0"}, {}]
sweep(f"c4 {size}", gbz, 2 * walks, reach, truth)
gbz.close()
c4_bench.cleanup(path)
if size != "full":
    s = S.Synth.chain(sites=333334, haplotypes=5000, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=42, extra=1)
    dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, flags=G.OPEN_EXTRACT)
    sweep("insertions", dev, np.arange(0, s.sequences, 2, dtype=np.uint64), [{"GBWT_HIP_GATHER_REACH": "0"}, {"GBWT_HIP_GATHER_REACH": "1"}, {"GBWT_HIP_GATHER_REACH": "8"}, {}])
    dev.close()
