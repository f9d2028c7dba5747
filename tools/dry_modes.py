#!/usr/bin/env python3
"""Headline kernel with the measurement switches of WalkArgs::debug (results are wrong by design with 1, 64, 128: no checks):
where the time of k_walk_direct goes."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
s = S.Synth.chain(sites=333334, haplotypes=5000, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=42)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
for label, env in (("default", {}), ("rows not stored (1)", {"GBWT_HIP_DEBUG_DRY_ROWS": "1"}), ("ring emptied unread (64)", {"GBWT_HIP_DEBUG_DRY_ROWS": "64"}),
                   ("rows into 1 MB (128)", {"GBWT_HIP_DEBUG_DRY_ROWS": "128"}), ("no look-ahead targets", {"GBWT_HIP_LOOKAHEAD_HOPS": "0"}),
                   ("plain stores (4)", {"GBWT_HIP_DEBUG_DRY_ROWS": "4"}), ("default again", {})):
    for k in ("GBWT_HIP_DEBUG_DRY_ROWS", "GBWT_HIP_LOOKAHEAD_HOPS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["GBWT_HIP_VMM"] = "0"
    dev.new_workspace()
    w = []
    for _ in range(12):
        dev.extract_device(ids)
        w.append(dev.last_kernel_ms()[0])
    print(f"{label:32s} k_walk_direct {np.mean(w[4:]):.3f} ms (min {min(w):.3f})", flush=True)
