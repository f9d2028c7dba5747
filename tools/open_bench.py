#!/usr/bin/env python3
"""Open time of an index (gbwt_hip_get_open_times) with checkpoint sampling and with the serial walk of every sequence,
and the extraction that follows: the one-shot flow of gbunzip (load, extract every path once, src/bin/gbunzip.rs:24-59).

  python tools/open_bench.py --sites 333334 --haplotypes 5000          # the headline index
  python tools/open_bench.py --sites 666667 --haplotypes 90            # config 4's shape on one contig (90 x 2 M nodes)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S

ap = argparse.ArgumentParser()
ap.add_argument("--sites", type=int, default=333334)
ap.add_argument("--haplotypes", type=int, default=5000)
ap.add_argument("--extra", type=int, default=0)
ap.add_argument("--indel-every", type=int, default=1)
ap.add_argument("--chop", type=int, default=1)
ap.add_argument("--passes", type=int, default=5)
ap.add_argument("--modes", default="checkpoint,serial", help="comma-separated: checkpoint, serial, or checkpoint:KNOB=VALUE:KNOB=VALUE (GBWT_HIP_ prefix added)")
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()

s = S.Synth.chain(args.sites, args.haplotypes, alleles=2, model=S.MOSAIC, seed=42, extra=args.extra, indel_every=args.indel_every, chop=args.chop)
hdr = dict(sequences=s.sequences, size=s.size, alphabet_offset=s.alphabet_offset, alphabet_size=s.alphabet_size)
data, starts = s.data(), s.starts()
truth = np.array([s.path_checksum(h) for h in range(s.paths)], dtype=np.uint64)
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
steps = (s.size - s.sequences) // 2
print(f"index: {args.haplotypes} haplotypes x {args.sites} sites (extra={args.extra}, every {args.indel_every}, chop={args.chop}): "
      f"{len(starts)} records, {len(data)} bytes, {steps} forward LF-steps", flush=True)
for mode in args.modes.split(","):
    for k in [k for k in os.environ if k.startswith("GBWT_HIP_") and k != "GBWT_HIP_LIB"]:
        del os.environ[k]
    if mode.startswith("serial"):
        os.environ["GBWT_HIP_SERIAL_SAMPLES"] = "1"
    for knob in mode.split(":")[1:]:
        k, v = knob.split("=")
        os.environ["GBWT_HIP_" + k] = v
    for rep in range(args.reps):
        t0 = time.perf_counter()
        dev = G.GBWT.from_records(data, starts, s.alphabet_offset, s.alphabet_size, s.sequences, s.size, bidirectional=True)
        t_open = time.perf_counter() - t0
        times = dev.open_times()
        t0 = time.perf_counter()
        out = dev.extract_device(ids)
        t_first = time.perf_counter() - t0
        walk = []
        for _ in range(args.passes):
            dev.extract_device(ids)
            walk.append(dev.last_kernel_ms()[0])
        assert int(out.total) == steps, (int(out.total), steps)
        assert np.array_equal(dev.path_sums(len(ids)), truth), "extracted paths differ from the generator's ground truth"
        cold = steps / (times["total_ms"] * 1e-3 + t_first)
        print(f"{mode:24s} open {t_open * 1e3:8.1f} ms wall (upload {times['upload_ms']:.1f}, samples {times['sample_ms']:.1f} ms; {times['samples']} samples, "
              f"{times['checkpoint_walkers']} walkers, {times['checkpoint_orphans']} hops ended at the cap); first extraction {t_first * 1e3:.1f} ms (workspace sizing included); "
              f"walk kernel {np.mean(walk):.3f} ms ({min(walk):.3f} min) = {steps / np.mean(walk) / 1e6:.0f} G LF-steps/s; "
              f"cold (open + first pass) {cold / 1e9:.1f} G LF-steps/s", flush=True)
        dev.close()
