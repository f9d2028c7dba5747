#!/usr/bin/env python3
"""What one rank of an N-GPU run does, on one GPU: the headline index, paths p = rank, rank + N, ... (bench.py --gpus N, strong scaling):
walk-kernel time and wall time per pass for N = 1, 2, 4, 8 -- the per-rank side of the scaling curve no box of rounds 1-4 could measure."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
s = S.Synth.chain(sites=333334, haplotypes=5000, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
print("interval", os.environ.get("GBWT_HIP_SAMPLE_INTERVAL", "default"), "open", {k: round(v, 2) if isinstance(v, float) else v for k, v in dev.open_times().items()}, flush=True)
base = None
for n in (1, 2, 4, 8):
    ids = 2 * np.arange(0, s.paths, n, dtype=np.uint64)
    for _ in range(5):
        dev.extract_device(ids)
    w, t0 = [], time.perf_counter()
    for _ in range(30):
        out = dev.extract_device(ids)
        w.append(dev.last_kernel_ms()[0])
    wall = (time.perf_counter() - t0) / 30 * 1e3
    k = float(np.mean(w))
    base = base or (k, wall)
    print(f"N={n}: {len(ids)} paths per rank: kernel {k:.3f} ms, wall {wall:.3f} ms per pass -> per-rank speed-up {base[0] / k:.2f} (kernel) {base[1] / wall:.2f} (wall) of {n}", flush=True)
