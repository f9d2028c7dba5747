#!/usr/bin/env python3
"""How the walk's iteration period depends on how many walkers share a CU: extractions of the first n paths of the
headline index (n = 64 -> 326 workgroups on 256 CUs ... n = 384 -> 1956 workgroups = about one round of 8 per CU).
Every walker runs 512 iterations of the two-step loop per 2048-node segment, so kernel time / 512 is the period.
With `half` the same workgroups run on XCDs 0-3 only: the load on the memory system stays, the load per CU doubles."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

s = S.Synth.chain(333334, 5000, alleles=2, model=S.MOSAIC, seed=42)
os.environ["GBWT_HIP_XCD_MAP"] = "0"
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
for dry in (0, 1):
    for half in (0, 16384):
        os.environ["GBWT_HIP_DEBUG_DRY_ROWS"] = str(dry + half)
        dev.new_workspace()   # the knobs are read when a workspace is created
        for n in (64, 128, 192, 256, 384, 768):
            ids = np.arange(0, 2 * n, 2, dtype=np.uint64)
            best = 1e9
            for _ in range(4):
                dev.extract_device(ids)
                best = min(best, dev.last_kernel_ms()[0])
            groups = -(-n // 64) * 326
            cus = 128 if half else 256
            rounds = max(1.0, groups / (8 * cus))
            print(f"dry={dry} {'XCDs 0-3' if half else 'all XCDs'} n={n:5d} workgroups={groups:6d} ({groups / cus:5.1f} per CU)  walk {best:7.3f} ms  -> {best * 2.4e6 / 512 / rounds:7.0f} cycles per iteration and round", flush=True)
