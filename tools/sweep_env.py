#!/usr/bin/env python3
"""Sweep of the environment knobs of the segmented extraction on one GPU (headline index by default).

Every config is a comma-separated list of NAME=VALUE (GBWT_HIP_ prefix implied); configs are separated by ';'.
Knobs read at open (SAMPLE_INTERVAL, LOOKAHEAD_HOPS) make the index reopen.  Every config is checked against the
generator's per-path checksums."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

OPEN_KNOBS = {"SAMPLE_INTERVAL", "LOOKAHEAD_HOPS", "SEQ_LEN", "TABLE_BYTES", "ORIENTATION_CHECK", "WALK_TABLES", "DEEP_TABLES", "CHAINS", "CHECKPOINT_GAP", "CHECKPOINT_CAP", "TWO_PASS_OPEN", "GATHER_LIMIT"}

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=333334)
ap.add_argument("--haplotypes", type=int, default=5000)
ap.add_argument("--model", default="mosaic")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--alleles", type=int, default=2)
ap.add_argument("--configs", default="")
ap.add_argument("--extra", type=int, default=0, help="allele 1 is an insertion of this many more nodes (walks leave lock step)")
ap.add_argument("--indel-every", type=int, default=1, help="... at every k-th site only")
ap.add_argument("--chop", type=int, default=1, help="every node a chain of this many nodes with consecutive ids")
args = ap.parse_args()

s = S.Synth.chain(args.sites, args.haplotypes, alleles=args.alleles, model=S.MOSAIC if args.model == "mosaic" else S.IID, seed=42, extra=args.extra, indel_every=args.indel_every, chop=args.chop)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
truth = np.array([s.path_checksum(h) for h in range(s.paths)], dtype=np.uint64)
steps = (s.size - s.sequences) // 2


def open_index():
    t0 = time.perf_counter()
    dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True)
    return dev, time.perf_counter() - t0


dev, open_s = None, 0.0
last_open = None
for cfg in [c for c in args.configs.split(";")] or [""]:
    knobs = dict(kv.split("=") for kv in cfg.split(",") if kv)
    for k in list(os.environ):
        if k.startswith("GBWT_HIP_"):
            del os.environ[k]
    for k, v in knobs.items():
        os.environ["GBWT_HIP_" + k] = v
    open_key = tuple(sorted((k, v) for k, v in knobs.items() if k in OPEN_KNOBS))
    if dev is None or open_key != last_open:
        dev = None
        dev, open_s = open_index()
        last_open = open_key
    else:
        dev.new_workspace()   # the extraction knobs are read when a workspace is created
    times, totals = [], []
    for _ in range(args.reps + 1):
        dev.extract_device(ids)
        times.append(dev.last_kernel_ms()[0])
        totals.append(dev.last_kernel_ms()[1])
    best, avg = min(times[1:]), float(np.mean(times[1:]))
    ok = np.array_equal(dev.path_sums(len(ids)), truth)
    print(f"{cfg or '(defaults)':60s} walk min {best:8.3f} avg {avg:8.3f} ms  all {min(totals[1:]):7.3f} ms  {steps / best / 1e6:8.1f} G steps/s  open {open_s:5.2f} s  ok={ok}", flush=True)
