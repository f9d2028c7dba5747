#!/usr/bin/env python3
"""One rank of eight, ten passes, for a rocprofv3 --pmc run: `parts` = stretch 4 of 8 of every path (gbwt_hip_extract_part_device), `paths` =
every 8th path whole.  tools/hbm_traffic.py turns the two counter passes into bytes per launch (profiles/r04_shard_traffic.txt)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
mode = sys.argv[1]
s = S.Synth.chain(sites=333334, haplotypes=5000, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
for _ in range(10):
    out = dev.extract_part_device(2 * np.arange(0, s.paths, dtype=np.uint64), 4, 8) if mode == "parts" else dev.extract_device(2 * np.arange(4, s.paths, 8, dtype=np.uint64))
print(mode, int(out.total), "LF-steps", dev.last_kernel_ms()[0], "ms", flush=True)
