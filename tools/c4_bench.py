#!/usr/bin/env python3
"""Config C4's shape on one GPU: Synth.genome -- 24 contigs x 20 graph components walked by random subsets of 90
haplotypes, ~32 000 ragged walks, 0.5 G LF-steps, 4.1 GB of P- and W-lines -- as gbunzip extracts it
(src/bin/gbunzip.rs:343-417): the P-lines of the generic sample, then the W-lines of all others.

Prints one JSON object: walk-only (kernel ms, LF-steps/s), walk + format with the text left in HBM (ms, text GB/s, bytes moved), and
optionally the whole file written to --out (/dev/shm/...).  bench.py imports run() for its `config4` object."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(contigs=24, fragments=20, haplotypes=90, sites=6000, passes=5, out="", device=0, keep=None):
    import gbwt_rs_amd as G
    from gbwt_rs_amd import synth as S
    t0 = time.perf_counter()
    g = S.Synth.genome(contigs=contigs, fragments=fragments, haplotypes=haplotypes, sites=sites, seed=42)
    gen_s = time.perf_counter() - t0
    tmpdir = tempfile.mkdtemp(prefix="gbwt_c4_")
    path = os.path.join(tmpdir, "c4.gbz")
    g.save(path, as_gbz=True)
    generic = np.array(g.generic_paths(), dtype=np.uint64)
    t0 = time.perf_counter()
    gbz = G.GBZ.load(path, device=device)
    open_ms = (time.perf_counter() - t0) * 1e3
    walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
    steps = (gbz.len() - gbz.sequences()) // 2
    res = {"workload": f"Synth.genome: {contigs} contigs x {fragments} components, {haplotypes} haplotypes, {sites} sites per component on average, seed 42: "
                       f"{g.paths} paths ({len(generic)} generic P-lines, {len(walks)} ragged W-lines), {steps} LF-steps",
           "paths": int(g.paths), "lf_steps": int(steps), "open_ms": open_ms, "generator_seconds": round(gen_s, 1)}
    # walk only: all forward sequences of the walks -> device CSR (the ragged batch: walker order computed per request)
    ids = 2 * walks
    for _ in range(2):
        o = gbz.extract_device(ids)
    wk, tot, wall = [], [], []
    for _ in range(passes):
        t0 = time.perf_counter()
        o = gbz.extract_device(ids)
        wall.append((time.perf_counter() - t0) * 1e3)
        w, t = gbz.last_kernel_ms()
        wk.append(w)
        tot.append(t)
    walk_steps = int(o.total)
    res["walk"] = {"kernel": "k_walk_direct", "kernel_ms": float(np.mean(wk)), "stream_ms": float(np.mean(tot)), "wall_ms": float(np.mean(wall)),
                   "lf_steps": walk_steps, "value_kernel": walk_steps / (np.mean(wk) * 1e-3), "value": walk_steps / (np.mean(wall) * 1e-3)}
    # walk + format, text left in HBM: ONE request for the P-lines, ONE for the W-lines
    def lines_pass():
        t0 = time.perf_counter()
        p = gbz.path_lines_device(generic, 0)
        p_total, (p_walk, p_fmt) = int(p.total), gbz.last_lines_ms()
        w = gbz.path_lines_device(walks, 1)
        wall_ms = (time.perf_counter() - t0) * 1e3
        w_walk, w_fmt = gbz.last_lines_ms()
        return wall_ms, p_total + int(w.total), p_walk + w_walk, p_fmt + w_fmt
    for _ in range(2):
        lines_pass()
        gbz.path_lines_device(walks[:1], 1)          # (another request in between: the next one is not answered from the cache)
    rows = []
    for _ in range(passes):
        rows.append(lines_pass())
        gbz.path_lines_device(walks[:1], 1)
    wall_ms, text, walk_ms, fmt_ms = (float(np.mean([r[k] for r in rows])) for k in range(4))
    text = int(text)
    moved = 4 * steps + 4 * steps * 2 + text       # rows written by the walk, read by the sizing pass and by the formatter, text written
    res["walk_format"] = {"ms": wall_ms, "walk_kernel_ms": walk_ms, "format_stream_ms": fmt_ms, "text_bytes": text, "text_GB_per_s": text / wall_ms / 1e6,
                          "value": steps / (wall_ms * 1e-3), "bytes_moved": moved, "achieved_GB_per_s": moved / wall_ms / 1e6,
                          "frac": moved / wall_ms / 1e6 / 8000.0,
                          "note": "bytes_moved = node ids written once by the walk (4 B/step), read by the sizing pass and by the formatter, + the text written; "
                                  "frac = that / wall time / 8 TB/s"}
    res["memory"] = gbz.memory_usage()
    if out:
        t0 = time.perf_counter()
        gbz.write_gfa(out)
        file_s = time.perf_counter() - t0
        size = os.path.getsize(out)
        res["whole_file"] = {"path": out, "bytes": size, "seconds": file_s, "GB_per_s": size / file_s / 1e9}
    if keep is not None:
        keep.update(gbz=gbz, synth=g, path=path, generic=generic, walks=walks)
    else:
        gbz.close()
        os.remove(path)
        os.rmdir(tmpdir)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--fragments", type=int, default=20)
    ap.add_argument("--haplotypes", type=int, default=90)
    ap.add_argument("--sites", type=int, default=6000)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    print(json.dumps(run(a.contigs, a.fragments, a.haplotypes, a.sites, a.passes, a.out)), flush=True)
