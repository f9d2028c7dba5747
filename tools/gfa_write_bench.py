#!/usr/bin/env python3
"""gbwt_hip_write_gfa (the whole file gbunzip writes) for a 20 000-site x 5 000-haplotype synthetic GBZ: bytes and wall time."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
s = S.Synth.chain(20000, 5000, alleles=2, model=S.MOSAIC, seed=42)
d = tempfile.mkdtemp()
p = os.path.join(d, "b.gbz"); s.save(p, as_gbz=True)
gbz = G.GBZ.load(p)
for rep in range(2):
    out = os.path.join(d, "o.gfa")
    t0 = time.perf_counter(); gbz.write_gfa(out); dt = time.perf_counter() - t0
    print(f"write_gfa: {os.path.getsize(out)} bytes in {dt:.2f} s = {os.path.getsize(out)/dt/1e9:.2f} GB/s")
