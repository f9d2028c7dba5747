#!/usr/bin/env python3
"""The open of the headline .gbz FILE (gbwt_hip_open_file_flags) phase by phase (GBWT_HIP_TRACE_OPEN=1), a few times, for the handle kinds a
caller can ask for, with the HIP runtime already started: where the 25 ms of round 5 go and what moves them (round 6)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
sites = int(sys.argv[1]) if len(sys.argv) > 1 else 333334
haps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
s = S.Synth.chain(sites=sites, haplotypes=haps, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=42)
path = os.path.join(tempfile.mkdtemp(prefix="gbwt_open_probe_"), "bench.gbz")
s.save(path, as_gbz=True)
tiny = S.Synth.chain(sites=8, haplotypes=4, alleles=2, model=S.MOSAIC, founders=2, switch_rate=0.1, seed=1)
t = G.GBWT.from_records(tiny.data(), tiny.starts(), tiny.alphabet_offset, tiny.alphabet_size, tiny.sequences, tiny.size, True)
t.sequences_csr(np.arange(tiny.sequences, dtype=np.uint64)); t.close()
ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
os.environ["GBWT_HIP_TRACE_OPEN"] = "1"
for label, flags, env in (("ALL", G.OPEN_ALL, {}), ("EXTRACT", G.OPEN_EXTRACT, {}), ("GFA", G.OPEN_GFA, {}), ("SEARCH", G.OPEN_SEARCH, {}), ("EXTRACT", G.OPEN_EXTRACT, {})):
    for k, v in env.items():
        os.environ[k] = v
    for rep in range(3):
        print(f"==== {label} #{rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        dev = G.GBZ.load(path, flags=flags)
        wall = (time.perf_counter() - t0) * 1e3
        ot = dev.open_times()
        if flags == G.OPEN_SEARCH:
            print(f"{label:20s} open {wall:7.2f} ms (parse {ot['parse_ms']:.2f}, upload {ot['upload_ms']:.2f})", flush=True)
            dev.close()
            continue
        t0 = time.perf_counter()
        out = dev.extract_device(ids)
        first = (time.perf_counter() - t0) * 1e3
        print(f"{label:20s} open {wall:7.2f} ms (parse {ot['parse_ms']:.2f}, upload {ot['upload_ms']:.2f}, samples {ot['sample_ms']:.2f}, line sizes {ot['line_sizes_ms']:.2f}); "
              f"first pass {first:.2f} ms, kernel {dev.last_kernel_ms()[0]:.2f}; cold {int(out.total) / (wall + first) / 1e6:.1f} G LF-steps/s", flush=True)
        dev.close()
    for k in env:
        del os.environ[k]
os.remove(path)
