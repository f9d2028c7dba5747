#!/usr/bin/env python3
"""Extraction from a graph whose haplotypes do NOT move in lock-step: a bubble chain in which allele 1 of every site is an
insertion (1 + EXTRA nodes instead of one; Synth.chain(extra=...)), so that the rows of a wave drift apart node by node and sit
on different records.  Every path is checked against the generator's allele matrix (checksums for all, node by node for a few);
the plain bubble chain (--extra 0) of the same shape is the comparison.  GBWT_HIP_* knobs pass through."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

ap = argparse.ArgumentParser()
ap.add_argument("--haplotypes", type=int, default=5000)
ap.add_argument("--sites", type=int, default=333334)
ap.add_argument("--founders", type=int, default=32)
ap.add_argument("--switch", type=float, default=2e-3)
ap.add_argument("--extra", default="0,1,3", help="comma-separated: extra nodes of allele 1 (0 = plain bubble chain)")
ap.add_argument("--indel-every", default="1", help="comma-separated: insertions at every k-th site only (the other sites are plain bubbles)")
ap.add_argument("--chop", type=int, default=1, help="every logical node is a chain of this many nodes with consecutive ids (most records unary, as in a GBZ built from a GFA with long segments)")
ap.add_argument("--repeats", type=int, default=5)
args = ap.parse_args()

cases = [(int(x), int(k)) for x in args.extra.split(",") for k in (args.indel_every.split(",") if int(x) else ["1"])]
for extra, every in cases:
    t0 = time.perf_counter()
    s = S.Synth.chain(args.sites, args.haplotypes, alleles=2, model=S.MOSAIC, founders=args.founders, switch_rate=args.switch, seed=42, extra=extra,
                      indel_every=every, chop=args.chop)
    build_s = time.perf_counter() - t0
    dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    nodes = (s.size - s.sequences) // 2
    walks, alls = [], []
    for _ in range(args.repeats):
        out = dev.extract_device(ids)
        ms = dev.last_kernel_ms()
        walks.append(ms[0]); alls.append(ms[1])
    ok = int(out.total) == nodes
    for k in (0, 1, len(ids) // 2, len(ids) - 1):
        ok = ok and np.array_equal(dev.copy_path(k), s.path(k))
    sums = dev.path_sums(len(ids))
    ok = ok and all(int(sums[k]) == s.path_checksum(k) for k in range(len(ids)))
    chopped = f", nodes chopped x {args.chop}" if args.chop > 1 else ""
    print(f"{args.haplotypes} haplotypes x {args.sites} sites, allele 1 = {1 + extra} node(s) at every {every}. site{chopped}: {nodes} nodes, generated in {build_s:.1f} s; "
          f"walk kernel {min(walks):.3f} ms (median {sorted(walks)[len(walks) // 2]:.3f}), whole pass {min(alls):.3f} ms = "
          f"{nodes / min(alls) / 1e6:.1f} G LF-steps/s  ok={ok}", flush=True)
    del dev, s
