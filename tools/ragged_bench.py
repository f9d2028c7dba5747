#!/usr/bin/env python3
"""Extraction of a path set with very different lengths: one long haplotype through a bubble chain and many short
walks over its first sites (the shape of a fragmented assembly).  Walk-kernel time and bit-exactness vs the input."""
import argparse
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=300000)
ap.add_argument("--long", type=int, default=4)
ap.add_argument("--short", type=int, default=50000)
ap.add_argument("--short-sites", type=int, default=40)
args = ap.parse_args()
rng = random.Random(5)


def walk(first, count):
    p = []
    for site in range(first, first + count):
        p.append(2 * (3 * site + 1))
        p.append(2 * (3 * site + 2 + (rng.random() < 0.3)))
    return p


paths = [walk(0, args.sites) for _ in range(args.long)] + [walk(rng.randrange(0, args.sites - args.short_sites), args.short_sites) for _ in range(args.short)]
s = S.Synth.from_paths(paths, bidirectional=True)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
nodes = sum(len(p) for p in paths)
best = None
for _ in range(4):
    out = dev.extract_device(ids)
    w = dev.last_kernel_ms()[0]
    best = w if best is None else min(best, w)
ok = int(out.total) == nodes and all(np.array_equal(dev.copy_path(k), np.array(paths[k], dtype=np.uint32)) for k in (0, args.long - 1, args.long, len(paths) - 1))
print(f"{args.long} paths of {2 * args.sites} nodes + {args.short} paths of {2 * args.short_sites}: {nodes} nodes, walk {best:.3f} ms, {nodes / best / 1e6:.1f} G LF-steps/s, ok={ok}")
