#!/usr/bin/env python3
"""Shader clock and power while the headline extraction runs back to back (rocm-smi sampled from a second thread)."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

s = S.Synth.chain(333334, 5000, alleles=2, model=S.MOSAIC, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            card = next(iter(json.loads(out).values()))
            samples.append((time.perf_counter(), {k: v for k, v in card.items() if "sclk" in k.lower() or "mclk" in k.lower() or "power" in k.lower() or "fclk" in k.lower()}))
        except Exception as e:  # noqa
            samples.append((time.perf_counter(), {"error": repr(e)}))
        time.sleep(0.05)


for dry in ("0", "1", "64"):
    os.environ["GBWT_HIP_DEBUG_DRY_ROWS"] = dry
    dev.new_workspace()   # the knobs are read when a workspace is created
    samples.clear()
    stop = False
    t = threading.Thread(target=sampler)
    t.start()
    time.sleep(0.5)
    t0 = time.perf_counter()
    walk = []
    while time.perf_counter() - t0 < 4.0:
        dev.extract_device(ids)
        walk.append(dev.last_kernel_ms()[0])
    t1 = time.perf_counter()
    time.sleep(0.5)
    stop = True
    t.join()
    print(f"dry={dry}: {len(walk)} passes, walk avg {np.mean(walk):.3f} ms (first 5: {[round(x, 3) for x in walk[:5]]}, last 5: {[round(x, 3) for x in walk[-5:]]})")
    for ts, v in samples:
        tag = "idle" if ts < t0 or ts > t1 else "busy"
        print(f"   {ts - t0:6.2f} s {tag}  {v}")
