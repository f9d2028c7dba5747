#!/usr/bin/env python3
"""The host-buffer form of the extraction on the headline index: gbwt_hip_extract (size query + extraction + device-to-host
copy of the node ids into a fresh numpy array).  The PCIe-inclusive rate DESIGN.md quotes; never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
s = S.Synth.chain(333334, 5000, alleles=2, model=S.MOSAIC, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
for rep in range(3):
    t0 = time.perf_counter()
    offsets, nodes = dev.sequences_csr(ids)
    dt = time.perf_counter() - t0
    print(f"gbwt_hip_extract into host buffers (size query + extraction + D2H of {nodes.nbytes / 1e9:.1f} GB): {dt:.2f} s = {len(nodes) / dt / 1e9:.1f} G LF-steps/s", flush=True)
