#!/usr/bin/env python3
"""A/B of the ways a large host-pointer query batch reaches the kernel (capi.hip: run_query): GBWT_HIP_QUERY_PIPELINE 0 = one piece over the
workspace stream, 1 = chunks through the copy lanes (upload / kernel / download of different chunks at once), 2 = one launch with both copies
through the lanes, 3 = only the copy back through the lanes; x piece size x copy threads.  Config 3's index and queries (tools/configs.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import configs as K  # noqa: E402

sites = int(sys.argv[1]) if len(sys.argv) > 1 else 1100000
keep = {}
K.search(sites=sites, passes=1, keep=keep)
dev, queries = keep["dev"], keep["queries"]
n, length = queries.shape
print(f"{n} queries of {length} nodes; one call moves {n * (8 * length + 25) / 1e6:.0f} MB (unidirectional), {n * (8 * length + 49) / 1e6:.0f} MB (bidirectional)")
ref = dev.search(queries)
for mode, piece, threads in [(0, 2048, 8)] + [(m, p, t) for m in (1, 2, 3) for p in (512, 2048, 8192) for t in (4, 8, 16)]:
    os.environ.update(GBWT_HIP_QUERY_PIPELINE=str(mode), GBWT_HIP_QUERY_PIECE_KIB=str(piece), GBWT_HIP_COPY_THREADS=str(threads))
    w = dev.another_workspace()
    row = []
    for fn in (lambda: w.search(queries), lambda: w.bd_search(queries, length // 2)):
        fn(), fn()
        t = []
        for _ in range(7):
            t0 = time.perf_counter()
            out = fn()
            t.append((time.perf_counter() - t0) * 1e3)
        row.append(float(np.median(t)))
    assert np.array_equal(w.search(queries)[0], ref[0])
    print(f"mode {mode} piece {piece:5d} KiB threads {threads:2d}: search {row[0]:6.2f} ms  bd_search {row[1]:6.2f} ms", flush=True)
    w.close()
