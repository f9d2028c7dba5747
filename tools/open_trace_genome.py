#!/usr/bin/env python3
"""Phases of the open (GBWT_HIP_TRACE_OPEN=1) of a config-4-shaped GBZ: Synth.genome with the defaults of tools/gfa_sharded.py."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GBWT_HIP_TRACE_OPEN"] = "1"
import gbwt_rs_amd as G
from gbwt_rs_amd import synth as S
s = S.Synth.genome(contigs=24, fragments=20, haplotypes=90, sites=6000, seed=42)
path = os.path.join(tempfile.mkdtemp(prefix="genome_"), "genome.gbz")
s.save(path, as_gbz=True)
print(f"{s.paths} paths, {s.sequences} sequences, {len(s.starts())} records, {len(s.data())} bytes of records, file {os.path.getsize(path)} bytes", flush=True)
tiny = G.GBZ.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "example.gbz"))
sys.stderr.write("---- the genome ----\n")
for rep in range(2):
    t0 = time.perf_counter()
    gbz = G.GBZ.load(path)
    print(f"open {1e3 * (time.perf_counter() - t0):.1f} ms", gbz.open_times(), flush=True)
    gbz.close()
os.remove(path)
