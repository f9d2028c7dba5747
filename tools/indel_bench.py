#!/usr/bin/env python3
"""Extraction from a graph whose haplotypes do NOT move in lock-step: a bubble chain in which one allele of every site is an
insertion (two nodes instead of one), so that the rows of a wave drift apart node by node and sit on different records.
Built with the brute-force builder (gbwt_synth_from_paths), checked against the input paths, timed next to the plain
bubble chain of the same size (GBWT_HIP_* knobs pass through)."""
import argparse
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

ap = argparse.ArgumentParser()
ap.add_argument("--haplotypes", type=int, default=1000)
ap.add_argument("--sites", type=int, default=10000)
ap.add_argument("--founders", type=int, default=32)
ap.add_argument("--switch", type=float, default=2e-3)
ap.add_argument("--insertion", type=int, default=1, help="extra nodes of allele 1 (0 = plain bubble chain)")
args = ap.parse_args()
rng = random.Random(11)

# mosaic alleles: founders carry per-site draws, haplotypes copy a founder and switch now and then
p_site = [rng.uniform(0.05, 0.95) for _ in range(args.sites)]
founders = [[rng.random() < p_site[s] for s in range(args.sites)] for _ in range(args.founders)]
stride = 3 + args.insertion
paths = []
for h in range(args.haplotypes):
    f = rng.randrange(args.founders)
    p = []
    for s in range(args.sites):
        if rng.random() < args.switch:
            f = rng.randrange(args.founders)
        p.append(2 * (stride * s + 1))
        if founders[f][s]:
            for k in range(1 + args.insertion):
                p.append(2 * (stride * s + 3 + k))
        else:
            p.append(2 * (stride * s + 2))
    paths.append(p)
nodes = sum(len(p) for p in paths)
t0 = time.perf_counter()
s = S.Synth.from_paths(paths, bidirectional=True)
build_s = time.perf_counter() - t0
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
best = None
for _ in range(4):
    out = dev.extract_device(ids)
    w = dev.last_kernel_ms()[0]
    best = w if best is None else min(best, w)
ok = int(out.total) == nodes and all(np.array_equal(dev.copy_path(k), np.array(paths[k], dtype=np.uint32)) for k in (0, 1, len(paths) // 2, len(paths) - 1))
sums = dev.path_sums(len(ids))
ok = ok and all(int(sums[k]) == sum(paths[k]) for k in range(len(paths)))
lens = sorted(len(p) for p in paths)
print(f"{args.haplotypes} haplotypes x {args.sites} sites, allele 1 = {1 + args.insertion} node(s): {nodes} nodes (rows of {lens[0]} .. {lens[-1]}), built in {build_s:.1f} s; "
      f"walk {best:.3f} ms = {nodes / best / 1e6:.1f} G LF-steps/s  ok={ok}", flush=True)
