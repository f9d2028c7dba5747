#!/usr/bin/env python3
"""Search throughput on one GPU, the shape of src/bin/benchmark.rs:124-169 (config C3 of SURVEY 8d): queries of
--len nodes taken from the index itself (random start position, extended with GBWT::forward, discarded when the
sequence ends early), then find + (len - 1) x extend in one launch, and the bidirectional form.  Prints kernel time
(HIP events) and wall time (with host staging).  Parity of these kernels is covered by tests/test_gpu_parity.py."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=100000)
ap.add_argument("--haplotypes", type=int, default=5008)
ap.add_argument("--model", default="mosaic")
ap.add_argument("--queries", type=int, default=1000000)
ap.add_argument("--len", type=int, default=10)
ap.add_argument("--seed", type=int, default=7)
args = ap.parse_args()

s = S.Synth.chain(args.sites, args.haplotypes, alleles=2, model=S.MOSAIC if args.model == "mosaic" else S.IID, seed=42)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
rng = np.random.default_rng(args.seed)
first, alphabet = s.alphabet_offset + 1, s.alphabet_size
# random (node, offset): node uniform over the alphabet, offset uniform in its record (find gives the length)
want = int(args.queries * 1.3) + 1024
nodes = rng.integers(first, alphabet, size=want, dtype=np.uint64)
states, ok = dev.find(nodes)
lens = (states["end"] - states["start"]).astype(np.uint64)
keep = ok & (lens > 0)
nodes, lens = nodes[keep], lens[keep]
offsets = (rng.random(nodes.size) * lens).astype(np.uint64)
pos = np.zeros(nodes.size, dtype=G.POS_DTYPE)
pos["node"], pos["offset"] = nodes, offsets
rows = [pos["node"].copy()]
alive = np.ones(nodes.size, dtype=bool)
for _ in range(args.len - 1):
    pos, ok = dev.forward(pos)
    alive &= ok & (pos["node"] != 0)
    rows.append(pos["node"].copy())
queries = np.stack(rows, axis=1)[alive][: args.queries].astype(np.uint64)
n = queries.shape[0]
print(f"index: {args.sites} sites x {args.haplotypes} haplotypes ({args.model}), {n} queries of {args.len} nodes")

for name, fn in (("find + extend", lambda: dev.search(queries)), ("bd_find + extend_forward/backward", lambda: dev.bd_search(queries, args.len // 2))):
    fn()
    best_k, best_w = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        out, ok = fn()
        w = (time.perf_counter() - t0) * 1e3
        k = dev.last_query_ms()
        best_k = k if best_k is None else min(best_k, k)
        best_w = w if best_w is None else min(best_w, w)
    steps = n * args.len
    print(f"{name:36s} kernel {best_k:8.3f} ms  {n / best_k / 1e3:8.1f} M queries/s  {steps / best_k / 1e6:7.2f} G steps/s   "
          f"wall {best_w:8.1f} ms   found {int(ok.sum())}/{n}")
