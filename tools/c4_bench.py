#!/usr/bin/env python3
"""BASELINE config 4: "HPRC-minigraph-cactus-scale GBZ, full GFA extraction sharded over 8 x MI355X with RCCL gather".

No real HPRC file exists offline: the stand-in is Synth.genome at the size SURVEY 8(d) states -- SIZES["full"]: 24 contigs x 20 graph
components walked by (nearly all of) 90 haplotypes = ~42 000 ragged walks over ~90 M nodes with labels of realistic length (1 .. 1 024 bp,
~40 bp on average; one contig whose W-line end coordinates pass 2^32), ~5 G LF-steps, ~50 GB of P- and W-lines -- extracted as gbunzip does
(src/bin/gbunzip.rs:343-417): the P-lines of the generic sample, then the W-lines of all others.  SIZES["small"] is the one-base-per-node
stand-in of rounds 3-4 (32 286 walks over 16 M nodes), kept as `config4_small`.

  run()          one GPU: walk only, walk + format (text left in HBM), optionally the whole file
  run_sharded()  N ranks (torchrun + RCCL, the loopback ranks of the test build, or gloo on a shared GPU): rank r formats ITS block of
                 path ids (contiguous blocks: a rank's walks lie in an N-th of the graph components, profiles/r04_c4_shard_probe.txt),
                 gbwt_hip_gather_lines puts the text in path order on rank 0, and the gathered text is compared with rank 0 formatting
                 everything alone.  bench.py --gpus N calls it on every rank (its `config4` object at N > 1).

Prints one JSON object when run as a program (N = 1)."""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIZES = {
    # SURVEY 8(d) / BASELINE.md 5: ~43 k paths, ~90 M nodes, realistic labels
    "full": dict(contigs=24, fragments=20, haplotypes=90, sites=50000, labels=1, min_walkers=0.95, wrap_contig=23),
    # rounds 3-4: 32 286 walks over 16 M one-base nodes
    "small": dict(contigs=24, fragments=20, haplotypes=90, sites=6000, labels=0, min_walkers=0.5, wrap_contig=None),
    # the loopback rehearsal of the N > 1 flow on one GPU (tests/test_gpu_dist.py): ~3 000 walks, a second to generate
    "medium": dict(contigs=8, fragments=6, haplotypes=64, sites=2500, labels=1, min_walkers=0.9, wrap_contig=7),
    # seconds on any box: the rehearsals of the N > 1 flow (tests, gloo)
    "tiny": dict(contigs=6, fragments=4, haplotypes=24, sites=300, labels=1, min_walkers=0.7, wrap_contig=5),
}


def describe(size, g, generic, walks, steps):
    p = SIZES[size]
    return (f"Synth.genome[{size}]: {p['contigs']} contigs x {p['fragments']} components, {p['haplotypes']} haplotypes, {p['sites']} sites per component on "
            f"average, {'labels 1..1024 bp' if p['labels'] else '1 bp labels'}, seed 42: {g.paths} paths ({len(generic)} generic P-lines, {len(walks)} ragged "
            f"W-lines) over {g.alphabet_size // 2 - 1} node ids, {steps} LF-steps")


def generate(size, path, threads=None):
    """The config's GBZ written to `path` (+ the ids of its generic paths next to it); returns the Synth (ground truth)."""
    from gbwt_rs_amd import synth as S
    p = SIZES[size]
    t0 = time.perf_counter()
    g = S.Synth.genome(contigs=p["contigs"], fragments=p["fragments"], haplotypes=p["haplotypes"], sites=p["sites"], seed=42, labels=p["labels"],
                       min_walkers=p["min_walkers"], wrap_contig=p["wrap_contig"], threads=threads or min(32, os.cpu_count() or 1))
    g.generator_seconds = time.perf_counter() - t0
    t0 = time.perf_counter()
    g.save(path + ".tmp", as_gbz=True)
    np.save(path + ".generic.npy", np.array(g.generic_paths(), dtype=np.uint64))
    os.replace(path + ".tmp", path)
    g.save_seconds = time.perf_counter() - t0
    return g


def run(size="small", passes=5, out="", device=0, keep=None, cpu_leg=None):
    import gbwt_rs_amd as G
    tmpdir = tempfile.mkdtemp(prefix="gbwt_c4_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(tmpdir, "c4.gbz")
    g = generate(size, path)
    generic = np.load(path + ".generic.npy")
    # The HIP runtime is started before the open is timed, as bench.py does for the headline (context, code objects, a tiny open, one 64 MB
    # pageable copy each way: what a process pays once) -- this function runs in a process of its own when bench.py calls it.
    import torch
    from gbwt_rs_amd import synth as S
    t0 = time.perf_counter()
    tiny = S.Synth.chain(sites=8, haplotypes=4, alleles=2, model=S.MOSAIC, founders=2, switch_rate=0.1, seed=1)
    tiny_dev = G.GBWT.from_records(tiny.data(), tiny.starts(), tiny.alphabet_offset, tiny.alphabet_size, tiny.sequences, tiny.size, True, device=device)
    tiny_dev.sequences_csr(np.arange(tiny.sequences, dtype=np.uint64))
    tiny_dev.close()
    torch.empty(64 << 20, dtype=torch.uint8).cuda(device).cpu()
    torch.cuda.synchronize()
    runtime_init_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    gbz = G.GBZ.load(path, device=device, flags=G.OPEN_GFA)     # GFA extraction only: no search structures (gbwt_hip_open_file_flags)
    open_ms = (time.perf_counter() - t0) * 1e3
    walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
    steps = (gbz.len() - gbz.sequences()) // 2
    res = {"workload": describe(size, g, generic, walks, steps), "size": size, "paths": int(g.paths), "lf_steps": int(steps), "open_ms": open_ms, "runtime_init_ms": runtime_init_ms,
           "gbz_bytes": os.path.getsize(path), "generator_seconds": round(g.generator_seconds, 1), "save_seconds": round(g.save_seconds, 1)}
    # walk + format, text left in HBM: ONE request for the P-lines, ONE for the W-lines
    def lines_pass():
        t0 = time.perf_counter()
        p = gbz.path_lines_device(generic, 0)
        p_total, (p_walk, p_fmt) = int(p.total), gbz.last_lines_ms()
        w = gbz.path_lines_device(walks, 1)
        wall_ms = (time.perf_counter() - t0) * 1e3
        w_walk, w_fmt = gbz.last_lines_ms()
        return wall_ms, p_total + int(w.total), p_walk + w_walk, p_fmt + w_fmt
    # The sizes of every line (token bytes per chunk, end coordinates) come from the index since round 6 -- one walk at open fills the line
    # cache (open_times: line_sizes_ms) -- so NO request sizes a line and the first request of a path costs what every later one does, as in
    # gbunzip's flow where every path is formatted exactly once (src/bin/gbunzip.rs:421-434).  What the very first request still pays on top
    # is the allocation of its workspace (rows + text: first_request_ms); `value` is the mean of the passes behind it.
    first = lines_pass()
    gbz.path_lines_device(walks[:1], 1)              # (another request in between: the next one is not answered from the workspace's last result)
    second = lines_pass()
    gbz.path_lines_device(walks[:1], 1)
    rows = []
    for _ in range(passes):
        rows.append(lines_pass())
        gbz.path_lines_device(walks[:1], 1)
    wall_ms, text, walk_ms, fmt_ms = (float(np.mean([r[k] for r in rows])) for k in range(4))
    text = int(text)
    moved = 4 * steps + 4 * steps + text           # rows written by the walk and read by the formatter, text written (line sizes: the index's line cache)
    times = gbz.open_times()
    res["line_sizes_ms"] = times.get("line_sizes_ms", 0.0)
    res["walk_format"] = {"ms": wall_ms, "walk_kernel_ms": walk_ms, "format_stream_ms": fmt_ms, "text_bytes": text, "text_GB_per_s": text / wall_ms / 1e6,
                          "value": steps / (wall_ms * 1e-3), "bytes_moved": moved, "achieved_GB_per_s": moved / wall_ms / 1e6,
                          "frac": moved / wall_ms / 1e6 / 8000.0, "first_request_ms": first[0], "second_request_ms": second[0],
                          "value_first_request": steps / (first[0] * 1e-3), "line_sizes_at_open_ms": times.get("line_sizes_ms", 0.0),
                          # the first request, split: the time between the HIP events around the walk and around the format on the stream, and the rest
                          # (the host: hipMalloc of the rows and the text, 73 GB at the stated size, launches).  The host's part is 1-3 ms.  The device's
                          # is the two kernels (33 ms) -- or seconds: memory that a process has released is cleared by the driver ON THE DEVICE before
                          # the first kernel may touch the allocation it went into, at 30-40 GB/s (tools/vram_first_touch_probe.py,
                          # profiles/r06_vram_first_touch.txt: 100 GB usable after 0.4 ms or after 0.5 / 2.2 / 3.3 / 5.2 s by what ran before): 1.8-2.4 s of
                          # a first request on a box in that state, none of it this library's work, all of it inside the events
                          "first_request_device_ms": first[2] + first[3], "first_request_host_ms": first[0] - first[2] - first[3],
                          "note": "bytes_moved = node ids written once by the walk (4 B/step) and read once by the formatter + the text written; no request sizes "
                                  "its lines (the index knows them since its open: line_sizes_at_open_ms, inside open_ms), so every pass formats every path "
                                  "as gbunzip does -- once, from nothing but the index; first_request_ms also holds the allocation of the workspace's rows and "
                                  "text buffers, and on the device whatever the driver runs before fresh memory may be touched (first_request_device_ms: 33 ms of "
                                  "kernels, or seconds when it first clears memory that a process released just before); frac = bytes_moved / wall time / 8 TB/s"}
    # walk only: all forward sequences of the walks -> device CSR (the ragged batch: walker order computed per request).  Behind the lines
    # passes: its third request rebuilds the rows from spread chunks (GBWT_HIP_VMM), and memory a process gives back is paid for by its NEXT
    # large allocation (profiles/r05_alloc_microbench.txt: hipMalloc of 16 GiB 0.2 ms, 2.5 s right after a hipFree of 48 GiB) -- medians
    ids = 2 * walks
    for _ in range(2):
        o = gbz.extract_device(ids)
    wk, tot, wall = [], [], []
    for _ in range(max(passes, 3)):
        t0 = time.perf_counter()
        o = gbz.extract_device(ids)
        wall.append((time.perf_counter() - t0) * 1e3)
        w, t = gbz.last_kernel_ms()
        wk.append(w)
        tot.append(t)
    walk_steps = int(o.total)
    res["walk"] = {"kernel": "k_walk_direct", "kernel_ms": float(np.median(wk)), "stream_ms": float(np.median(tot)), "wall_ms": float(np.median(wall)),
                   "lf_steps": walk_steps, "value_kernel": walk_steps / (np.median(wk) * 1e-3), "value": walk_steps / (np.median(wall) * 1e-3)}
    res["memory"] = gbz.memory_usage()
    if cpu_leg is not None:       # bench.py's cpu_baseline leg: the oracle formats a bounded sample of these W-lines, compared byte for byte with the device's
        res["cpu_baseline"] = cpu_leg(path, gbz, walks)
    if out:
        t0 = time.perf_counter()
        gbz.write_gfa(out)
        file_s = time.perf_counter() - t0
        size_b = os.path.getsize(out)
        res["whole_file"] = {"path": out, "bytes": size_b, "seconds": file_s, "GB_per_s": size_b / file_s / 1e9}
    if keep is not None:
        keep.update(gbz=gbz, synth=g, path=path, generic=generic, walks=walks, tmpdir=tmpdir)
    else:
        gbz.close()
        cleanup(path)
    return res


def cleanup(path):
    for f in (path, path + ".generic.npy", path + ".tmp"):
        if os.path.exists(f):
            os.remove(f)
    try:
        os.rmdir(os.path.dirname(path))
    except OSError:
        pass


def device_bytes_equal(a, b, piece=1 << 28):
    """Two uint8 device tensors hold the same bytes (compared in pieces: no 50 GB temporary)."""
    import torch
    if a.numel() != b.numel():
        return False
    return all(bool(torch.equal(a[lo:lo + piece], b[lo:lo + piece])) for lo in range(0, a.numel(), piece))


def write_at(fd, data, at, threads=8, piece=32 << 20):
    """`data` (a numpy uint8 array) into the open file at byte `at`, by a few threads with positional writes."""
    from concurrent.futures import ThreadPoolExecutor
    view = memoryview(data)

    def one(lo):
        done = 0
        chunk = view[lo:lo + piece]
        while done < len(chunk):
            done += os.pwrite(fd, chunk[done:], at + lo + done)

    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(one, range(0, len(data), piece)))


def run_sharded(gbz, generic, walks, rank, world, comm, barrier, device, passes=3, check=True, torch_gather=None, file_path=None, allgather=None):
    """The N-rank flow of config 4 on an index that every rank has opened (`gbz`: this rank's handle / workspace).  Collective: every rank
    calls it.  `comm`: a gbwt_rs_amd.dist.Comm (RCCL behind the C ABI, or the loopback ranks of the test build); None = gather through
    `torch_gather` (lengths, text) -> (offsets, text) on rank 0 (the torch.distributed form: what the gloo rehearsal takes).
    `barrier()` meets all ranks.  Returns the `config4` object on rank 0, per-rank numbers elsewhere."""
    import torch
    from gbwt_rs_amd import dist as D
    dev = torch.device("cuda", device)
    lo, hi = D.shard_bounds(len(walks), rank, world)
    mine = walks[lo:hi]                                   # a block of consecutive path ids = an N-th of the graph components
    p_lo, p_hi = D.shard_bounds(len(generic), rank, world)
    my_generic = generic[p_lo:p_hi]
    other = walks[:1] if len(walks) else walks

    def lines_pass():
        t0 = time.perf_counter()
        p = gbz.path_lines_device(my_generic, 0)
        p_total, (p_walk, p_fmt) = int(p.total), gbz.last_lines_ms() if len(my_generic) else (0.0, 0.0)
        w = gbz.path_lines_device(mine, 1)
        wall_ms = (time.perf_counter() - t0) * 1e3
        w_walk, w_fmt = gbz.last_lines_ms() if len(mine) else (0.0, 0.0)
        return wall_ms, p_total + int(w.total), p_walk + w_walk, p_fmt + w_fmt

    lines_pass()
    gbz.path_lines_device(other, 1)                       # (another request in between: the next one is not answered from the cache)
    barrier()
    t0 = time.perf_counter()
    rows = []
    for _ in range(passes):
        rows.append(lines_pass())
        gbz.path_lines_device(other, 1)
    barrier()
    loop_ms = (time.perf_counter() - t0) * 1e3 / passes   # (with the in-between requests: an upper bound of a pass, the same clock on every rank)
    wall_ms, text, walk_ms, fmt_ms = (float(np.mean([r[k] for r in rows])) for k in range(4))
    my = {"rank": rank, "walks": int(len(mine)), "p_lines": int(len(my_generic)), "wall_ms": wall_ms, "walk_kernel_ms": walk_ms, "format_stream_ms": fmt_ms,
          "text_bytes": int(text), "loop_ms": loop_ms}

    # ---- the one exchange: the W-lines of every rank in path order on rank 0 (the P-lines the same way: they are few) --------------------
    def gather(ids, mode):
        lines = gbz.path_lines_device(ids, mode)
        barrier()
        t0 = time.perf_counter()
        if comm is not None:
            got = comm.gather_lines(gbz, root=0, interleaved=False)
            stats = comm.last()
            off, txt = D.lines_tensors(got, dev) if rank == 0 else (None, None)
        else:
            l_off, l_txt = D.lines_tensors(lines, dev)
            off, txt = torch_gather(l_off, l_txt)
            stats = None
        barrier()
        return off, txt, (time.perf_counter() - t0) * 1e3, stats

    _, _, _, _ = gather(mine, 1)                           # untimed: connections, buffers
    w_off, w_txt, gather_ms, stats = gather(mine, 1)
    my["gather_ms"] = gather_ms
    my["comm"] = stats
    res = my
    if rank == 0:
        total_text = int(w_txt.numel())
        sha_w = None
        if check:                                             # (w_txt: the communicator's buffer, valid until its next gather -- the P-lines below)
            alone = gbz.another_workspace()
            a_lines = alone.path_lines_device(walks, 1)
            a_off, a_txt = D.lines_tensors(a_lines, dev)
            assert int(a_lines.total) == total_text, (int(a_lines.total), total_text)
            assert device_bytes_equal(w_txt.to(dev), a_txt), "the gathered W-lines differ from one rank formatting alone"
            if comm is not None:                              # (the torch.distributed form sends a rank's block of lines as ONE row: no per-line offsets)
                assert bool(torch.equal(w_off.to(dev), a_off)), "line offsets of the gathered W-lines differ"
            head = min(total_text, 64 << 20)
            sha_w = hashlib.sha256(w_txt[:head].cpu().numpy().tobytes()).hexdigest()
            assert sha_w == hashlib.sha256(a_txt[:head].cpu().numpy().tobytes()).hexdigest()
            alone.close()
        res = {"text_bytes": total_text, "gather_ms": gather_ms, "gather_GB_per_s": total_text / gather_ms / 1e6,
               "check": None if not check else "gathered W-lines == rank 0 formatting all walks alone: every byte compared on the device" + (", line offsets equal, " if comm is not None else ", ") +
                                               f"sha256 of the first {min(total_text, 64 << 20)} bytes {sha_w}"}
    # ---- the other way to the file: nobody gathers text -- every rank copies ITS lines to the host and writes them at
    # their place in the one file (the offsets come from an all-gather of the ranks' byte counts: 8 bytes per rank instead of 51 GB to
    # rank 0 and one writer there).  `allgather(value) -> [value of rank 0, ...]`.
    if file_path is not None and allgather is not None:
        side = gbz.another_workspace()                        # the few P-lines: a workspace of their own, the W-lines stay in the main one
        lines = gbz.path_lines_device(mine, 1)
        p_bytes = allgather(int(side.path_lines_device(my_generic, 0).total) if len(my_generic) else 0)
        w_bytes = allgather(int(lines.total))
        p_total = sum(p_bytes)
        # (every rank must take the same way: the file system has room for the file, or nobody writes)
        import shutil
        room = min(allgather(int(shutil.disk_usage(os.path.dirname(file_path) or ".").free)))
        if room < 1.1 * (p_total + sum(w_bytes)) + (1 << 30):
            if rank == 0:
                res["sharded_file"] = {"skipped": f"{room} bytes free under {os.path.dirname(file_path)}, the file needs {p_total + sum(w_bytes)}"}
            side.close()
            file_path = None
    if file_path is not None and allgather is not None:
        if rank == 0:
            with open(file_path, "wb") as f:
                f.truncate(p_total + sum(w_bytes))
        barrier()
        t0 = time.perf_counter()
        fd = os.open(file_path, os.O_WRONLY)
        try:
            if len(my_generic):
                write_at(fd, side.path_lines_array(my_generic, 0), sum(p_bytes[:rank]))
            text = gbz.path_lines_array(mine, 1)              # (the request of a moment ago: copied out of the workspace, not formatted again)
            write_at(fd, text, p_total + sum(w_bytes[:rank]))
        finally:
            os.close(fd)
        barrier()
        file_s = time.perf_counter() - t0
        side.close()
        my["file_seconds"] = file_s
        if rank == 0:
            size = os.path.getsize(file_path)
            assert size == p_total + sum(w_bytes), (size, p_total, sum(w_bytes))
            res["sharded_file"] = {"bytes": size, "seconds": file_s, "GB_per_s": size / file_s / 1e9,
                                   "note": "every rank writes its own P- and W-lines at their offsets in ONE file (offsets from an all-gather of byte counts): no text "
                                           "travels between GPUs; compare with gather_ms + one writer on rank 0"}
            if check:                                         # the file against rank 0 formatting alone, at seeded places (and whole when it is small)
                alone = gbz.another_workspace()
                a_txt = D.lines_tensors(alone.path_lines_device(walks, 1), dev)[1]
                whole = np.memmap(file_path, dtype=np.uint8, mode="r")
                gen = np.random.default_rng(3)
                window = min(1 << 20, int(a_txt.numel()))
                for lo in [0, int(a_txt.numel()) - window] + [int(x) for x in gen.integers(0, max(1, int(a_txt.numel()) - window), 62)]:
                    assert np.array_equal(whole[p_total + lo:p_total + lo + window], a_txt[lo:lo + window].cpu().numpy()), ("sharded file differs at", lo)
                assert bytes(whole[:p_total]) == alone.path_lines(generic, 0), "P-lines of the sharded file differ"
                del whole
                alone.close()
        barrier()
    p_off, p_txt, p_gather_ms, _ = gather(my_generic, 0)
    if rank == 0 and check:
        alone = gbz.another_workspace()
        a_off, a_txt = D.lines_tensors(alone.path_lines_device(generic, 0), dev)
        assert device_bytes_equal(p_txt.to(dev), a_txt), "the gathered P-lines differ from one rank formatting alone"
        alone.close()
    if rank == 0:
        res["p_lines_gather_ms"] = p_gather_ms
        res["p_text_bytes"] = int(p_txt.numel())
    return res, my


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", choices=sorted(SIZES), default="small")
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    print(json.dumps(run(a.size, a.passes, a.out)), flush=True)
