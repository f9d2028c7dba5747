#!/usr/bin/env python3
"""A rank of N that walks ALL paths over its N-th of the way (gbwt_hip_extract_part_device, bench.py --shard parts), against a rank that
walks every N-th path whole (tools/shard_probe.py): kernel and wall ms per pass on one GPU, for the first, a middle and the last rank."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
s = S.Synth.chain(sites=333334, haplotypes=5000, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
ids = 2 * np.arange(0, s.paths, dtype=np.uint64)
print("stride", os.environ.get("GBWT_HIP_SAMPLE_STRIDE", "auto"), flush=True)
for n in (1, 2, 4, 8):
    for r in sorted({0, n // 2, n - 1}):
        for _ in range(5):
            dev.extract_part_device(ids, r, n)
        w, t0 = [], time.perf_counter()
        for _ in range(30):
            out = dev.extract_part_device(ids, r, n)
            w.append(dev.last_kernel_ms()[0])
        wall = (time.perf_counter() - t0) / 30 * 1e3
        print(f"N={n} rank {r}: {int(out.total)} LF-steps: kernel {np.mean(w):.3f} ms, wall {wall:.3f} ms per pass", flush=True)
