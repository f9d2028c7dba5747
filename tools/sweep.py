#!/usr/bin/env python3
"""Tuning sweep of the extraction kernel on one GPU: walk-kernel time for several (mode, paths_per_wave, small_record)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=50000)
ap.add_argument("--haplotypes", type=int, default=5000)
ap.add_argument("--model", default="mosaic")
ap.add_argument("--alleles", type=int, default=2)
ap.add_argument("--configs", default="0:64:16,0:16:16,0:4:16,2:64:16,2:8:16,1:64:16")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--subset", type=int, default=0, help="walk only the first N paths")
ap.add_argument("--dup", type=int, default=1, help="every DUP consecutive lanes walk the same path (memory-coalescing experiment)")
args = ap.parse_args()

s = S.Synth.chain(args.sites, args.haplotypes, alleles=args.alleles, model=S.MOSAIC if args.model == "mosaic" else S.IID, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
if args.subset:
    ids = ids[:args.subset]
if args.dup > 1:
    ids = np.repeat(ids[::args.dup], args.dup)[:len(ids)]
truth = np.array([s.path_checksum(int(i) // 2) for i in ids], dtype=np.uint64)
steps = (s.size - s.sequences) // 2 * len(ids) // s.paths
for cfg in args.configs.split(","):
    mode, p, small = (int(x) for x in cfg.split(":"))
    dev.tune(mode, p, small)
    best = None
    for _ in range(args.reps):
        dev.extract_device(ids)
        w, t = dev.last_kernel_ms()
        best = w if best is None else min(best, w)
    ok = np.array_equal(dev.path_sums(len(ids)), truth)
    print(f"mode={mode} paths_per_wave={p:2d} small_record={small:3d}  walk {best:9.2f} ms  {steps / best / 1e6:9.1f} M steps/s  "
          f"{best * 1e3 / (2 * args.sites):7.3f} us/step  ok={ok}", flush=True)
