#!/usr/bin/env python3
"""Does the time of a pass depend on where the allocations land?  Fresh process per mode:
  plain            open, extract
  hold:<GB>        a torch allocation of that size is made first and kept
  freed:<GB>       ... made first and freed (the caching allocator is emptied) before the open
  reopen:<k>       the index is opened and closed k times before the measured open"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
s = S.Synth.chain(333334, 5000, alleles=2, model=S.MOSAIC, seed=42)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
keep = None
if mode.startswith("hold:") or mode.startswith("freed:"):
    import torch
    keep = torch.empty(int(float(mode.split(":")[1]) * (1 << 30)), dtype=torch.uint8, device="cuda")
    if mode.startswith("freed:"):
        del keep
        keep = None
        torch.cuda.empty_cache()


def opened():
    return G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)


if mode.startswith("reopen:"):
    for _ in range(int(mode.split(":")[1])):
        d = opened()
        d.extract_device(ids)
        del d
import time
dev = opened()
t0 = time.perf_counter()
dev.extract_device(ids)
first_s = time.perf_counter() - t0
times = []
for _ in range(5):
    dev.extract_device(ids)
    times.append(dev.last_kernel_ms()[0])
print(f"{mode:12s} walk min {min(times[1:]):.3f} avg {np.mean(times[1:]):.3f} ms   first extraction (sizes the workspace) {first_s * 1e3:.0f} ms   GBWT_HIP_VMM={os.environ.get('GBWT_HIP_VMM', '(default)')}", flush=True)
