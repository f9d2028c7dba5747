#!/usr/bin/env python3
"""Config 4's shape, the per-rank side of an N-GPU run on one GPU: the W-lines of a rank's share of the walks (walk + format, text left in
HBM), for walks dealt to ranks by path id mod N, in blocks of consecutive path ids, and in blocks of the walks ordered by the
graph component they lie in (a rank then touches an N-th of the index)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import c4_bench
keep = {}
res = c4_bench.run(passes=1, keep=keep)
gbz, g, walks = keep["gbz"], keep["synth"], keep["walks"]
print("whole batch: walk + format", round(res["walk_format"]["ms"], 3), "ms;", len(walks), "walks", flush=True)
# the component of a walk = the record its first node has (walks of one component start within its node range)
first = np.array([int(gbz.start(int(2 * p))[0]) if gbz.start(int(2 * p)) else 0 for p in walks[:0]], dtype=np.int64)   # (kept empty: see below)
off, nodes = gbz.sequences_csr(2 * walks[:])
first = nodes[off[:-1].astype(np.int64)].astype(np.int64)
by_component = walks[np.argsort(first, kind="stable")]

def measure(sub):
    for _ in range(2):
        gbz.path_lines_device(sub, 1); gbz.path_lines_device(walks[:1], 1)
    wall, dev = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        gbz.path_lines_device(sub, 1)
        wall.append((time.perf_counter() - t0) * 1e3)
        dev.append(sum(gbz.last_lines_ms()))
        gbz.path_lines_device(walks[:1], 1)
    return float(np.mean(dev)), float(np.mean(wall))

base = measure(walks)
print(f"N=1: walk + format {base[0]:.3f} ms on the stream, {base[1]:.3f} ms wall", flush=True)
for n in (2, 4, 8):
    for name, order, pick in (("p mod N", walks, lambda a, r: a[r::n]), ("blocks of path ids", walks, lambda a, r: a[len(a) * r // n:len(a) * (r + 1) // n]),
                              ("blocks by component", by_component, lambda a, r: np.sort(a[len(a) * r // n:len(a) * (r + 1) // n]))):
        rows = [measure(pick(order, r)) for r in sorted({0, n // 2, n - 1})]
        worst = max(rows, key=lambda x: x[1])
        print(f"N={n} {name:20s}: slowest of ranks 0 / {n // 2} / {n - 1}: {worst[0]:.3f} ms stream, {worst[1]:.3f} ms wall -> speed-up {base[1] / worst[1]:.2f} of {n}", flush=True)
