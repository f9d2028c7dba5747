#!/usr/bin/env python3
"""bench.py -- LF-steps/s of batched path extraction on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over the whole batch: extract every forward sequence of the
rank's shard (what gbunzip extracts, src/bin/gbunzip.rs:447-532) into a device-resident CSR.
Workload: the headline config, bubble chain 333,334 sites x 5,000 haplotypes (1,000,002 nodes,
3.33 G forward LF-steps, seed 42, mosaic).  At N > 1 (default --scaling strong) the index
is replicated on every rank and path p is walked by rank p mod N; no collective inside the timed
region; afterwards the whole CSR is gathered on rank 0 over RCCL.  --scaling weak: every rank holds
its own contig of the same shape (seed 42 + rank).

Next to the steady-state `value` the line carries the one-shot flow (open_ms, sample_walk_ms,
first_pass_ms, value_cold), the same passes without sequence samples (value_unsampled) and a second
workload whose rows do not move in lock step (secondary) and BASELINE config 5 (high_degree).

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.

`--gpus N` with N > 1 outside a torchrun launch starts the N ranks itself (a child `python -m torch.distributed.run`,
before this process has touched a GPU) and passes their output through; under torchrun (RANK / WORLD_SIZE set) it is
one of the ranks.  A world size that contradicts --gpus is an error, not something to paper over.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

_PROCESS_START = time.perf_counter()

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a pass takes 4.5 ms; the first few after the workspace has been sized run 2-4 % slower (720 G LF-steps/s with 1 + 3 passes,
    # 737 with 3 + 5, 748 with 5 + 10 on one box), so the defaults leave them to the warm-up
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sites", type=int, default=333334)
    ap.add_argument("--haplotypes", type=int, default=5000)
    ap.add_argument("--model", choices=["mosaic", "iid"], default="mosaic")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target wall time of the timed cpu_baseline walk (the untimed parity walk behind it takes as long again)")
    ap.add_argument("--bytes-sample", type=int, default=16, help="paths used for the reference-pattern bytes pass")
    ap.add_argument("--gather-paths", type=int, default=32, help="N > 1, weak scaling: paths per rank whose W-lines go through the final RCCL gather")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = ONE index (seed 42) replicated on every rank, path p walked by rank p mod N, whole CSR gathered on rank 0 "
                         "(SURVEY 8e; default); weak = every rank its own contig (seed 42 + rank)")
    ap.add_argument("--shard", choices=["parts", "paths"], default="parts",
                    help="N > 1, strong scaling: parts = every rank walks EVERY path over its N-th of the way (rows cut at sequence samples: gbwt_hip_extract_part_device; "
                         "default since round 4), paths = path p walked whole by rank p mod N (rounds 1-3)")
    ap.add_argument("--no-extras", action="store_true", help="skip value_unsampled and the other BASELINE configs (secondary = insertion chain, high_degree = config 5, search = config 3, config4)")
    ap.add_argument("--no-search", action="store_true", help="skip the config 3 search object (its 1.1 M-site index takes half a minute to generate)")
    ap.add_argument("--no-config4", action="store_true", help="skip the config 4 object")
    ap.add_argument("--c4-size", choices=["full", "small", "medium", "tiny"], default="full",
                    help="config 4's stand-in (tools/c4_bench.py: SIZES): full = the size SURVEY 8(d) states (~42 000 walks over ~90 M nodes, labels of 1..1024 bp); "
                         "small / tiny for rehearsals")
    return ap.parse_args()


def cpu_baseline(index_path, n_paths, target_seconds):
    """The oracle (a port of the reference algorithm with its per-step costs), timed on this host's cores.
    Only this leg of bench.py touches oracle/.  The timed walk keeps one (length, sum, order-dependent hash) per path, so the leg is also
    a parity check of the sampled paths at full size: returns (object for the line, path ids walked, lengths, sums, hashes)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import subprocess
    import oracle_lib as O
    kind_note = "generic x86-64 build"
    if O._lib is None:
        try:  # the reference builds with target-cpu=native (.cargo/config.toml:1-2): rebuild for this host
            subprocess.check_call(["make", "-C", O.ORACLE_DIR, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            native = os.path.join(O.ORACLE_DIR, "_native", "liboracle.so")
            if os.path.exists(native):
                O.LIB_OVERRIDE, kind_note = native, "-O3 -march=native build"
        except Exception:
            pass
    threads = min(os.cpu_count() or 1, 64)  # gbunzip caps its pool at 64 (src/bin/gbunzip.rs:94-95)
    oracle = O.OracleGBZ(index_path).gbwt()
    # calibration: one path per thread
    ids = np.arange(0, 2 * min(threads, n_paths), 2, dtype=np.uint64)
    t0 = time.perf_counter()
    steps = oracle.extract_timed(ids, threads)
    dt = time.perf_counter() - t0
    rate = steps / dt
    per_path = steps / len(ids)
    # (one path per thread walks a third faster per step than a sample that keeps every thread busy for seconds: 0.7 of the calibrated
    # rate puts the timed walk near its target instead of 1.4-1.5 times over it)
    want = int(max(len(ids), min(n_paths, 0.7 * target_seconds * rate / per_path)))
    ids = np.arange(0, 2 * want, 2, dtype=np.uint64)
    # TIMED: the plain walk (SequenceIter::next and nothing else per step, src/gbwt.rs:557-568).  UNTIMED, behind it: the same paths again with
    # a length, a node sum and an order-dependent hash kept per path -- the parity pass.  (Round 5 timed the checksum walk: a sum and a
    # multiply-add per LF-step that the reference's iterator does not do; `checksum_walk_ratio` says what that cost.)
    t0 = time.perf_counter()
    steps = oracle.extract_timed(ids, threads)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    c_steps, lengths, sums, hashes = oracle.extract_checksums(ids, threads)
    dt_checks = time.perf_counter() - t0
    assert c_steps == steps
    return ({"value": steps / dt, "unit": "LF-steps/s", "cores": threads, "kind": "port", "checksum_walk_ratio": round(dt_checks / dt, 3),
             "sample": f"{want} of {n_paths} forward paths ({steps} LF-steps, {dt:.1f} s wall, {kind_note}, "
                       f"pthread pool pulling path ids like gbunzip's rayon par_iter); behind the timed walk the same paths are walked again, untimed "
                       f"({dt_checks:.1f} s), keeping every path's length, node sum and order-dependent hash, which are compared with the GPU's"}, ids // 2, lengths, sums, hashes)


def _oracle_of_synth(s):
    """(cpu_baseline legs only) The CPU oracle over a generated index held in memory."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    bwt = O.OracleBWT.from_parts(bytes(s.data()), s.starts())
    return O.OracleGBWT.from_bwt(bwt, s.sequences, s.size, s.alphabet_offset, s.alphabet_size, s.bidirectional)


def cpu_leg_extraction(seconds):
    """cpu_baseline leg of an extraction config (tools/configs.py calls it with the generated index and the length / sum / hash of every row
    of its last GPU extraction of all forward sequences): the oracle walks a bounded sample of the same paths, timed, and must agree."""
    def leg(s, lens, sums, hashes):
        threads = min(os.cpu_count() or 1, 64)
        oracle = _oracle_of_synth(s)
        n_paths = len(lens)
        ids = np.arange(0, 2 * min(threads, n_paths), 2, dtype=np.uint64)
        t0 = time.perf_counter()
        steps = oracle.extract_timed(ids, threads)
        rate, per_path = steps / (time.perf_counter() - t0), steps / len(ids)
        want = int(max(len(ids), min(n_paths, seconds * rate / max(per_path, 1))))
        ids = np.arange(0, 2 * want, 2, dtype=np.uint64)
        t0 = time.perf_counter()
        steps = oracle.extract_timed(ids, threads)                      # timed: the plain walk; the parity pass behind it is not
        dt = time.perf_counter() - t0
        _, o_lens, o_sums, o_hashes = oracle.extract_checksums(ids, threads)
        assert np.array_equal(lens[:want], o_lens) and np.array_equal(sums[:want], o_sums) and np.array_equal(hashes[:want], o_hashes), \
            "the extracted paths differ from the oracle's walk"
        return {"value": steps / dt, "unit": "LF-steps/s", "cores": threads, "kind": "port", "parity_checked_paths": want,
                "sample": f"{want} of {n_paths} forward paths ({steps} LF-steps, {dt:.1f} s wall), lengths / sums / hashes equal to the GPU's"}
    return leg


def cpu_leg_search(seconds):
    """cpu_baseline leg of config 3: the oracle's find + extend loop (src/bin/benchmark.rs:155-169) over a bounded sample of the same queries
    on min(cores, 64) threads, in the reference's own units (src/internal.rs:58-65: time per query, time per node); final states compared."""
    def leg(s, queries, states, ok):
        threads = min(os.cpu_count() or 1, 64)
        oracle = _oracle_of_synth(s)
        n, length = queries.shape
        t0 = time.perf_counter()
        oracle.search_batch(queries[:2048 * threads // 8 + 256], threads)
        rate = (2048 * threads // 8 + 256) / (time.perf_counter() - t0)
        want = int(max(1024, min(n, seconds * rate)))
        t0 = time.perf_counter()
        o_st, o_ok = oracle.search_batch(queries[:want], threads)
        dt = time.perf_counter() - t0
        got = np.stack([states["node"], states["start"], states["end"]], axis=1)[:want]
        assert np.array_equal(ok[:want], o_ok) and np.array_equal(got[o_ok], o_st[o_ok]), "final states differ from the oracle's"
        return {"value": want / dt, "unit": "queries/s", "us_per_query": dt * 1e6 / want, "ns_per_node": dt * 1e9 / (want * length), "cores": threads,
                "kind": "port", "parity_checked_queries": want,
                "sample": f"the first {want} of {n} queries, find + {length - 1} x extend ({dt:.1f} s wall), every final state equal to the GPU's"}
    return leg


def cpu_leg_lines(seconds):
    """cpu_baseline leg of config 4: the oracle's path_to_w_line (src/bin/gbunzip.rs:495-550: walk + format) over a bounded sample of the walks
    on min(cores, 64) threads pulling path ids like gbunzip's rayon pool; every line compared byte for byte with the device's text."""
    def leg(gbz_path, gbz, walks):
        import torch
        from concurrent.futures import ThreadPoolExecutor
        from gbwt_rs_amd import dist as D
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        threads = min(os.cpu_count() or 1, 64)
        t0 = time.perf_counter()
        oracle = O.OracleGBZ(gbz_path)
        load_s = time.perf_counter() - t0
        pick = np.random.default_rng(11).choice(len(walks), min(len(walks), 64 * threads), replace=False)
        with ThreadPoolExecutor(threads) as pool:
            t0 = time.perf_counter()
            lines = list(pool.map(lambda k: oracle.path_lines([int(walks[k])], 1), pick[:threads]))
            rate = sum(len(x) for x in lines) / (time.perf_counter() - t0)                      # text bytes per second, all threads
            want = int(max(threads, min(len(pick), seconds * rate / max(1, np.mean([len(x) for x in lines])))))
            pick = pick[:want]
            t0 = time.perf_counter()
            lines = list(pool.map(lambda k: oracle.path_lines([int(walks[k])], 1), pick))
            dt = time.perf_counter() - t0
        device = torch.device("cuda", gbz._device)
        got = gbz.path_lines_device(walks, 1)
        off, text = D.lines_tensors(got, device)
        off = off.cpu().numpy()
        nodes = 0
        for k, line in zip(pick, lines):
            assert text[int(off[k]):int(off[k + 1])].cpu().numpy().tobytes() == line, f"W-line of walk {int(walks[k])} differs from the oracle's"
            nodes += line.count(b">") + line.count(b"<")
        return {"value": nodes / dt, "unit": "LF-steps/s", "text_GB_per_s": sum(len(x) for x in lines) / dt / 1e9, "cores": threads, "kind": "port",
                "parity_checked_lines": int(len(pick)), "oracle_load_seconds": round(load_s, 1),
                "sample": f"{len(pick)} of {len(walks)} W-lines ({nodes} LF-steps, {dt:.1f} s wall): walk + format per path as gbunzip's workers do, every line equal to the device's"}
    return leg


def algorithmic_bytes(index_path, n_paths, sample):
    """Exact algorithmic bytes W = sum(H + P + 4) over a sample of paths (untimed oracle pass)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    oracle = O.OracleGBZ(index_path).gbwt()
    ids = np.arange(0, 2 * min(sample, n_paths), 2, dtype=np.uint64)
    total, steps = oracle.algorithmic_bytes(ids)
    return total / steps, int(steps)


def final_gather(args, index, out, s, rank, world, local_rank, backend, comm_device, by_parts, strong, n_paths, all_steps, my_paths, barrier, dist, torch):
    """The one exchange of the job: the extracted rows of every rank on rank 0, in path order, over RCCL point-to-point sends (one group,
    every peer over its own xGMI link).  Strong scaling: the WHOLE CSR (13.3 GB / N per rank) through the C ABI (gbwt_hip_comm_*: RCCL
    called by the library, rows placed by kernels on rank 0); where the library cannot make a communicator (no RCCL: GBWT_HIP_UNSUPPORTED;
    the gloo rehearsal on a shared GPU) ALL ranks agree on the torch.distributed form.  Weak scaling: the W-lines of a bounded sample of
    every rank's paths.  Wrong rows are an error, never a fallback."""
    import gbwt_rs_amd as G
    from gbwt_rs_amd import dist as D
    device = torch.device("cuda", local_rank)
    payload_note = "the whole CSR: node ids (u32) + row lengths of every rank's shard, gathered on rank 0 in path order"
    if not strong:
        sample = np.arange(min(args.gather_paths, n_paths), dtype=np.uint64)
        lines = index.path_lines_device(sample, 1)
        line_off, text = D.lines_tensors(lines, device)
        mine = text.clone()
        if comm_device == "cpu":
            line_off, text = line_off.cpu(), text.cpu()
        sizes = torch.tensor([float(text.numel())], dtype=torch.float64, device=comm_device)
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
        barrier()
        tg = time.perf_counter()
        g_off, g_text = D.gather_lines(line_off, text, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        info = {"ms": gather_ms, "bytes": int(sizes.item()), "payload": f"W-lines of {len(sample)} paths of every rank, formatted on the device", "backend": backend}
        if rank == 0:
            assert g_text.numel() == int(sizes.item()) and int(g_off[1]) == mine.numel()
            assert torch.equal(g_text[:mine.numel()].to(mine.device), mine), "rank 0's own lines changed on the way"
            info["GB_per_s"] = int(sizes.item()) / 1e9 / (gather_ms * 1e-3)
        return info
    # every rank learns whether EVERY rank has a communicator before anybody enters a collective of either kind
    comm, why = None, None
    try:
        if comm_device != "cuda":
            raise RuntimeError("not an RCCL run (BENCH_DIST_BACKEND)")
        comm = D.Comm(rank, world, local_rank)           # (a failure of rank 0 to make the id is raised on every rank: dist.Comm)
    except (G.GbwtHipError, RuntimeError) as e:
        why = repr(e)[:200]
    agreed = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64, device=comm_device)
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
    if float(agreed.item()) == 1.0:
        layout = D.GATHER_PARTS if by_parts else D.GATHER_INTERLEAVED
        comm.gather_rows(index, root=0, layout=layout)          # untimed: connections, buffers
        barrier()
        tg = time.perf_counter()
        got = comm.gather_rows(index, root=0, layout=layout)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        info = {"ms": gather_ms, "backend": backend, "via": "gbwt_hip_comm (C ABI, RCCL)", "rank_stats": comm.last(), "payload": payload_note}
        if rank == 0:
            assert int(got.total) == int(all_steps) and int(got.n) == n_paths, (int(got.total), all_steps)
            g_off, g_nodes = D.paths_tensors(got, device)
            for p_id in list(range(min(world, n_paths))) + list(range(max(0, n_paths - world), n_paths)):   # first and last row of every rank
                row = g_nodes[int(g_off[p_id]):int(g_off[p_id + 1])].cpu().numpy().astype(np.uint32)
                assert np.array_equal(row, s.path(p_id)), f"row {p_id} changed on the way"
            info["bytes"] = 4 * int(got.total) + 8 * n_paths
            info["GB_per_s"] = info["bytes"] / 1e9 / (gather_ms * 1e-3)
        comm.close()
        return info
    if comm is not None:
        comm.close()
    offsets, nodes = D.paths_tensors(out, device)
    lengths = (offsets[1:] - offsets[:-1]).clone()
    payload = nodes.clone()                                     # (rows may be mapped from spread chunks: staged for the send)
    if comm_device == "cpu":
        lengths, payload = lengths.cpu(), payload.cpu()
    barrier()
    tg = time.perf_counter()
    len_parts, val_parts = D.gather_parts(lengths, payload, dst=0)
    barrier()
    gather_ms = (time.perf_counter() - tg) * 1e3
    info = {"ms": gather_ms, "backend": backend, "via": "torch.distributed batch_isend_irecv (no gbwt_hip_comm on some rank; this rank: " + str(why) + ")", "payload": payload_note}
    if rank == 0:
        got_n = sum(int(p.numel()) for p in val_parts)
        assert got_n == int(all_steps), (got_n, all_steps)
        if by_parts:                               # the stretches of every row, joined in rank order
            j_off, j_nodes = D.join_row_parts(len_parts, val_parts)
            for p_id in (0, n_paths // 2, n_paths - 1):
                row = j_nodes[int(j_off[p_id]):int(j_off[p_id + 1])].cpu().numpy().astype(np.uint32)
                assert np.array_equal(row, s.path(p_id)), f"row {p_id} changed on the way"
        for r in range(world if not by_parts else 0):   # first and last row of every rank's part against the generator
            paths_r = shard_paths(n_paths, r, world)
            ends = torch.cumsum(len_parts[r], 0)
            for k in (0, len(paths_r) - 1):
                lo = int(ends[k - 1]) if k else 0
                row = val_parts[r][lo:int(ends[k])].cpu().numpy().astype(np.uint32)
                assert np.array_equal(row, s.path(int(paths_r[k]))), f"row {k} of rank {r} changed on the way"
        info["bytes"] = 4 * got_n + 8 * n_paths
        info["GB_per_s"] = info["bytes"] / 1e9 / (gather_ms * 1e-3)
    return info


_C4_FILES = []          # what config4_sharded has put under /dev/shm (rank 0): removed by main() when the step is abandoned and its own `finally` never runs


def config4_sharded(args, rank, world, local_rank, backend, comm_device, barrier, dist, torch):
    """BASELINE config 4 as it is named: full GFA extraction sharded over the N GPUs with the RCCL gather (tools/c4_bench.py: run_sharded).
    Rank 0 generates the GBZ once (/dev/shm), every rank opens it (the index is replicated), formats the lines of its block of path ids
    and the text is gathered on rank 0; the gathered text is compared with rank 0 formatting alone.  Collective: every rank calls it;
    returns the object for the line on rank 0."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import c4_bench
    import gbwt_rs_amd as G
    from gbwt_rs_amd import dist as D
    box = [None]
    if rank == 0:
        try:                                        # (a generator that fails on rank 0 is told to everybody: they wait in the broadcast below)
            base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
            box[0] = os.path.join(tempfile.mkdtemp(prefix="gbwt_bench_c4_", dir=base), "c4.gbz")
            _C4_FILES.append(box[0])
            g = c4_bench.generate(args.c4_size, box[0])
            gen = {"generator_seconds": round(g.generator_seconds, 1), "save_seconds": round(g.save_seconds, 1), "paths": int(g.paths)}
            del g
        except Exception as e:  # noqa: BLE001
            box[0] = RuntimeError(f"config 4: the generator failed on rank 0: {e!r}"[:400])
    dist.broadcast_object_list(box, src=0)
    if isinstance(box[0], Exception):
        raise box[0]
    path = box[0]

    def agree(ok, what):
        """Every rank learns whether EVERY rank got through a step that can fail locally, BEFORE anybody enters the next collective: a rank
        that raised would leave its peers waiting in it (with RCCL until the watchdog aborts them, and the line with them)."""
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=comm_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) != 1.0:
            raise RuntimeError(f"config 4: {what} failed on some rank" + ("" if ok else " (this one)"))

    gbz = None
    try:
        failure = None
        try:
            generic = np.load(path + ".generic.npy")
            t0 = time.perf_counter()
            gbz = G.GBZ.load(path, device=local_rank, flags=G.OPEN_GFA)     # GFA extraction only: no search structures on any of the N replicas
            open_ms = (time.perf_counter() - t0) * 1e3
        except Exception as e:  # noqa: BLE001
            failure = e
        agree(failure is None, "opening the index" + (f": {failure!r}"[:300] if failure is not None else ""))
        walks = np.setdiff1d(np.arange(gbz.paths(), dtype=np.uint64), generic)
        steps = (gbz.len() - gbz.sequences()) // 2
        comm, why = None, None
        try:
            if comm_device != "cuda":
                raise RuntimeError("not an RCCL run (BENCH_DIST_BACKEND)")
            comm = D.Comm(rank, world, local_rank)
        except (G.GbwtHipError, RuntimeError) as e:
            why = repr(e)[:200]
        agreed = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64, device=comm_device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if float(agreed.item()) != 1.0 and comm is not None:
            comm.close()
            comm = None

        def torch_gather(off, text):                       # the torch.distributed form (gloo: CPU tensors)
            if comm_device == "cpu":
                off, text = off.cpu(), text.cpu()
            return D.gather_lines(off, text, dst=0)

        def allgather(value):
            got = [None] * world
            dist.all_gather_object(got, value)
            return got

        res, my = c4_bench.run_sharded(gbz, generic, walks, rank, world, comm, barrier, local_rank, passes=3, torch_gather=torch_gather,
                                       file_path=path + ".lines.gfa", allgather=allgather)
        everyone = [None] * world
        dist.all_gather_object(everyone, my)
        if comm is not None:
            comm.close()
        gbz.close()
        if rank != 0:
            return None
        slowest = max(r["loop_ms"] for r in everyone)
        res.update({"workload": f"BASELINE config 4, {args.c4_size}: {gen['paths']} paths, {steps} LF-steps, P- and W-lines; rank r formats its block of path ids, text "
                                f"gathered on rank 0 ({'gbwt_hip_gather_lines (C ABI, RCCL)' if comm is not None else 'torch.distributed: ' + str(why)})",
                    "size": args.c4_size, "lf_steps": int(steps), "n_gpus": world, "open_ms_rank0": open_ms, "ranks": everyone, "ms_per_pass": slowest,
                    "value": steps / (slowest * 1e-3), "value_incl_gather": steps / ((slowest + res["gather_ms"]) * 1e-3), "unit": "LF-steps/s", **gen})
        return res
    finally:
        try:
            barrier()                              # (nobody deletes the file while a rank still reads it; a barrier that fails must not mask what was raised)
        except Exception:  # noqa: BLE001
            pass
        if rank == 0:
            if os.path.exists(path + ".lines.gfa"):
                os.remove(path + ".lines.gfa")
            c4_bench.cleanup(path)


def launch_ranks(args):
    """--gpus N > 1 without a launcher: run the ranks as children of this (GPU-free) process and exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this host driver
    return subprocess.call(cmd, env=env)


def source_fingerprint():
    """sha256 over the kernel sources + the GBWT_HIP_* knobs in force: what a PMC profile must have been taken with for its
    numbers to be quoted next to this run (tools/hbm_traffic.py stores the same value)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "gbwt_rs_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "gbwt_rs_amd", "csrc", "*.cpp"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    for k in sorted(os.environ):
        if k.startswith("GBWT_HIP_") and k != "GBWT_HIP_LIB":
            h.update(f"{k}={os.environ[k]};".encode())
    return h.hexdigest()[:16]


def lookup_traffic(workload_key, kernel=None):
    """HBM bytes per launch of `kernel` (default: the profile's first kernel) from a rocprofv3 PMC profile under profiles/ that was taken
    with THESE kernel sources and knobs on THIS workload (tools/hbm_traffic.py stores both); (None, why) otherwise."""
    import glob
    fingerprint = source_fingerprint()
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), reverse=True):
        try:
            tj = json.load(open(tpath))
        except (OSError, ValueError):
            continue
        if tj.get("source_fingerprint") != fingerprint or tj.get("workload_key") != workload_key:
            continue
        entry = tj if kernel is None else tj.get("kernels", {}).get(kernel)
        if entry is None:
            continue
        factor = tj.get("fetch_factor", 2.0)
        return entry["traffic_bytes_per_launch"], (f"profiles/{os.path.basename(tpath)} (same kernel sources and knobs, fingerprint {fingerprint}): "
                                                   f"{factor:g} x FETCH_SIZE ({'gfx950 correction for wide coalesced loads' if factor == 2.0 else 'scattered 64-byte requests: counted exactly, profiles/r05_fetch_calibration.txt'}) "
                                                   "+ WRITE_SIZE, separate --pmc passes")
    return None, "no PMC profile of this build / workload under profiles/"


def config_roofline(obj, key, definition, kernel=None):
    """The roofline object of a secondary workload: algorithmic bytes / kernel time against the HBM peak, and the measured traffic
    (of `kernel` alone where the profile holds several and the object is about one of them)."""
    seconds = obj["kernel_ms"] * 1e-3
    achieved = obj["algorithmic_bytes"] / seconds / 1e9
    traffic, source = lookup_traffic(key, kernel)
    obj["roofline"] = {"bound": "hbm", "definition": definition, "kernel": obj["kernel"], "kernel_ms": obj["kernel_ms"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                       "traffic_frac": None if traffic is None else traffic / seconds / 1e9 / HBM_PEAK_GBS, "traffic_source": source}
    return obj


def shard_paths(n_paths, rank, world):
    """Path ids of one rank: p -> GPU p mod G (interleaved, because neighbouring ids correlate by contig and length)."""
    return np.arange(rank, n_paths, world, dtype=np.uint64)


def timed_passes(index, ids, passes, part=0, parts=1):
    """`passes` extractions of `ids`: (walk-kernel ms, everything-on-the-stream ms) per pass from the workspace's HIP events."""
    walk, total, out = [], [], None
    for _ in range(passes):
        out = index.extract_part_device(ids, part, parts) if parts > 1 else index.extract_device(ids)
        w, t = index.last_kernel_ms()
        walk.append(w)
        total.append(t)
    return out, walk, total


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    strong = args.scaling == "strong"

    import torch
    dist = None
    # BENCH_DIST_BACKEND=gloo + BENCH_SHARE_GPU=1 exist only to rehearse the N > 1 flow on a 1-GPU box
    # (all ranks on cuda:0, reductions on CPU tensors); the driver's runs use RCCL, one GPU per rank.
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        local_rank = 0
    comm_device = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        import datetime
        # (rank 0 generates config 4's GBZ for half a minute while the others wait for its path.  Longer than the guards below: with RCCL the process
        # group's watchdog ABORTS a process whose collective times out -- the guarded steps must be abandoned by their guard first, so that the line survives)
        patience = datetime.timedelta(minutes=45)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=patience)
        else:
            dist.init_process_group(backend=backend, timeout=patience)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: gbwt_rs_amd has no CPU fallback")

    import gbwt_rs_amd as G
    from gbwt_rs_amd import synth as S

    model = S.MOSAIC if args.model == "mosaic" else S.IID
    # strong scaling: ONE path set -- the headline index, seed 42, replicated on every rank -- whose paths are dealt
    # p -> rank p mod G; weak scaling: every rank has a contig of its own (seed 42 + rank) and walks all of it
    seed = args.seed if strong else args.seed + rank
    t0 = time.perf_counter()
    s = S.Synth.chain(sites=args.sites, haplotypes=args.haplotypes, alleles=2, model=model, founders=32, switch_rate=2e-3, seed=seed)
    gen_s = time.perf_counter() - t0
    tmpdir = tempfile.mkdtemp(prefix=f"gbwt_bench_r{rank}_")
    index_path = os.path.join(tmpdir, "bench.gbz")
    s.save(index_path, as_gbz=True)  # same .gbz for the GPU path and the CPU baseline

    # The HIP runtime (context, code objects) is started by a throw-away open of a tiny index, so that `open_ms` below is the open
    # of THIS index and not of the process; what a one-shot tool pays on top is reported as runtime_init_ms.
    t0 = time.perf_counter()
    tiny = S.Synth.chain(sites=8, haplotypes=4, alleles=2, model=model, founders=2, switch_rate=0.1, seed=1)
    tiny_dev = G.GBWT.from_records(tiny.data(), tiny.starts(), tiny.alphabet_offset, tiny.alphabet_size, tiny.sequences, tiny.size, True, device=local_rank)
    tiny_dev.sequences_csr(np.arange(tiny.sequences, dtype=np.uint64))
    tiny_dev.close()
    # ... and the runtime's path for LARGE pageable copies, which it sets up on first use (the first 60 MB host-to-device copy of a process
    # takes 31 ms, every later one 1.1 ms: profiles/r06_upload_probe.txt; the first open of a file in a process paid 8-10 ms of it in its
    # parse, profiles/r06_open_probe.txt): one 64 MB round trip of a throw-away buffer, counted with the runtime's start like the rest
    scratch = torch.empty(64 << 20, dtype=torch.uint8)
    scratch.cuda(local_rank).cpu()
    del scratch
    torch.cuda.synchronize()
    runtime_init_ms = (time.perf_counter() - t0) * 1e3

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the one-shot flow of gbunzip (src/bin/gbunzip.rs:24-59): load, extract every path once
    # The handle is opened for what this job calls -- extraction (gbwt_hip_open_file_flags: GBWT_HIP_OPEN_EXTRACT; weak scaling also formats a
    # sample of W-lines for its gather) -- as a caller of the library would: the search structures and the GFA tables of a handle opened for
    # everything (7 ms of line sizes, 1.1 GB) are built by the configs that use them (`search`, `config4`), not by this one.
    open_flags = G.OPEN_EXTRACT if strong else G.OPEN_ALL
    t0 = time.perf_counter()
    index = G.GBZ.load(index_path, device=local_rank, flags=open_flags)
    open_ms = (time.perf_counter() - t0) * 1e3
    open_times = index.open_times()
    n_paths = index.paths()
    # N > 1, strong scaling: every rank walks EVERY path over ITS N-th of the way (--shard parts; rows are cut where their walkers start
    # anyway, at sequence samples) -- a rank then reads an N-th of the index and keeps whole waves on every record; --shard paths is the
    # scheme of rounds 1-3, path p whole on rank p mod N (an eighth of the batch 0.88 ms there, 0.53-0.60 ms as parts: profiles/r04_shard_probe.txt)
    by_parts = strong and world > 1 and args.shard == "parts"
    parts = world if by_parts else 1
    my_paths = np.arange(n_paths, dtype=np.uint64) if (by_parts or not strong) else shard_paths(n_paths, rank, world)
    ids = 2 * my_paths
    extract = (lambda: index.extract_part_device(ids, rank, parts)) if by_parts else (lambda: index.extract_device(ids))
    t0 = time.perf_counter()
    out = extract()
    first_pass_ms = (time.perf_counter() - t0) * 1e3
    first_walk_ms = index.last_kernel_ms()[0]
    path_len = (index.len() - index.sequences()) // 2 // n_paths if n_paths else 0   # every path of this generator visits every site
    expected_steps = len(my_paths) * path_len                 # (by parts: of all ranks together, checked below)
    steps_done = int(out.total)

    # ---- steady state: the index resident, the same batch again and again
    for _ in range(args.warmup):
        extract()
    # The timed region is K x 4 ms: one pause of the interpreter's cycle collector over the generator's objects would be a tenth of it
    import gc
    gc.collect()
    gc.disable()
    try:
        barrier()
        t0 = time.perf_counter()
        out, walk_ms, total_ms = timed_passes(index, ids, args.steps, rank, parts)
        barrier()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    assert int(out.total) == steps_done and (by_parts or steps_done == expected_steps), (int(out.total), steps_done, expected_steps)

    # untimed: full-size check of the last extraction against the generator's ground truth
    sums = index.path_sums(len(ids))
    hashes = index.path_hashes(len(ids))
    row_lens = np.diff(index.last_offsets(len(ids)))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:      # (the generator's own walk of every path, in C without the interpreter lock: 5 000 x 667 k nodes)
        truth = np.array(list(pool.map(s.path_checksum, [int(p) for p in my_paths])), dtype=np.uint64)
    part_from = None
    if by_parts:
        # a row's stretches add up to the row: the per-row sums of all ranks together are the generator's, and this rank's stretch of a row
        # is the row from where the ranks before it stopped
        off_mine = index.last_offsets(len(ids))
        lens = torch.from_numpy(np.diff(off_mine).astype(np.int64)).to(comm_device)
        all_lens = [torch.zeros_like(lens) for _ in range(world)]
        dist.all_gather(all_lens, lens)
        part_from = sum((x.cpu().numpy() for x in all_lens[:rank]), np.zeros(len(ids), dtype=np.int64))
        row_len = sum(x.cpu().numpy() for x in all_lens)
        assert np.all(row_len == path_len), "the stretches of a row do not add up to its length"
        t_sums = torch.from_numpy(sums.astype(np.int64)).to(comm_device)      # (sums mod 2^64: int64 wraps the same way)
        dist.all_reduce(t_sums, op=dist.ReduceOp.SUM)
        assert np.array_equal(t_sums.cpu().numpy().astype(np.uint64), truth), "extracted paths differ from the generator's ground truth"
        for k in (0, len(ids) // 2, len(ids) - 1):
            lo = int(part_from[k])
            assert np.array_equal(index.copy_path(k), s.path(int(my_paths[k]))[lo:lo + int(lens[k])]), f"stretch {rank} of row {k}"
    else:
        assert np.array_equal(sums, truth), "extracted paths differ from the generator's ground truth"
        for k in (0, len(ids) // 2, len(ids) - 1):
            assert np.array_equal(index.copy_path(k), s.path(int(my_paths[k])))

    gather_info = None
    other_cut = None
    c4_sharded = None
    abandoned = []                                 # guarded steps of the N > 1 flow that did not come back (below)
    failed_checks = []                             # ... and those whose result failed its check
    cold_ms = open_ms + first_pass_ms              # the one-shot flow of the slowest rank
    rank_kernel_ms = [float(np.mean(walk_ms))]
    if dist is not None:
        t = torch.tensor([elapsed, cold_ms], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, cold_ms = float(t[0].item()), float(t[1].item())
        tot = torch.tensor([steps_done], dtype=torch.float64, device=comm_device)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        all_steps = float(tot.item())
        per_rank = torch.zeros(world, dtype=torch.float64, device=comm_device)
        dist.all_gather_into_tensor(per_rank, torch.tensor([float(np.mean(walk_ms))], dtype=torch.float64, device=comm_device))
        rank_kernel_ms = [float(x) for x in per_rank.cpu().tolist()]
        if strong:
            # ... and the SAME batch under the other cut (north_star: "the path set shards ... across the 8 GPUs" = --shard paths; the default
            # is --shard parts, every rank its stretch of every row): the same K timed passes between the same barriers
            o_parts = 1 if by_parts else world
            o_paths = shard_paths(n_paths, rank, world) if by_parts else np.arange(n_paths, dtype=np.uint64)
            o_ids = 2 * o_paths
            o_extract = (lambda: index.extract_device(o_ids)) if by_parts else (lambda: index.extract_part_device(o_ids, rank, o_parts))
            for _ in range(args.warmup):
                o_extract()
            barrier()
            t0 = time.perf_counter()
            o_out, o_walk, _ = timed_passes(index, o_ids, args.steps, rank, o_parts)
            barrier()
            o_elapsed = time.perf_counter() - t0
            t = torch.tensor([o_elapsed, float(np.mean(o_walk))], dtype=torch.float64, device=comm_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            o_tot = torch.tensor([float(int(o_out.total))], dtype=torch.float64, device=comm_device)
            dist.all_reduce(o_tot, op=dist.ReduceOp.SUM)
            assert float(o_tot.item()) == all_steps, "the two cuts of the batch do not hold the same LF-steps"
            other_cut = {"shard": "paths" if by_parts else "parts", "value": all_steps * args.steps / float(t[0].item()), "ms_per_step": float(t[0].item()) / args.steps * 1e3,
                         "slowest_kernel_ms": float(t[1].item()),
                         "sharding": ("path p walked whole by rank p mod G (SURVEY 8e; north_star's partition)" if by_parts else
                                      "every rank walks stretch r of EVERY path (gbwt_hip_extract_part_device)")}
        # FROM HERE ON NOTHING MAY TAKE THE LINE WITH IT.  What follows -- the gather of the rows through the library's own RCCL communicator, config 4
        # sharded over the ranks -- has run between loopback ranks, over gloo and at world size 1, never between GPUs: each step runs in a thread
        # that the main thread waits for a bounded time (BENCH_GUARD_SECONDS, default 600 per step).  A step that does not come back is
        # ABANDONED: its object says so, the steps behind it are skipped (the collectives are in an unknown state), rank 0 prints the line
        # with everything measured above, and the ranks leave without another barrier.  A step whose RESULT fails its check (wrong rows) is listed
        # under `errors` on the line -- loud, but it does not take the measurements above it, which were checked on their own, with it.
        guard_s = float(os.environ.get("BENCH_GUARD_SECONDS", "600"))

        def bounded(what, fn):
            import threading
            box = {}

            def run():
                try:
                    torch.cuda.set_device(local_rank)
                    box["value"] = fn()
                except AssertionError as e:
                    box["assertion"] = e
                except Exception as e:  # noqa: BLE001    (a transport that fails -- RCCL, the process group -- must not take the measurement with it)
                    box["error"] = e

            t = threading.Thread(target=run, daemon=True, name=what)
            t.start()
            t.join(guard_s)
            if t.is_alive():
                abandoned.append(what)
                return {"error": f"{what}: no answer within {guard_s:.0f} s on rank {rank}: abandoned (the ranks leave without a barrier)", "rank": rank}
            if "assertion" in box:
                # a check of the step's RESULT failed (gathered rows that differ, a sharded file that differs): not a fallback and not silent -- the
                # object says so in capitals, the line lists it under `errors` -- but the measurements above it were checked on their own and stay
                failed_checks.append(what)
                return {"error": "CHECK FAILED: " + repr(box["assertion"])[:500], "rank": rank}
            if "error" in box:
                return {"error": repr(box["error"])[:500], "rank": rank}
            return box["value"]

        # The one exchange of the job (outside the timed region, src/bin/gbunzip.rs:421-434: the writer's mutex)
        gather_info = bounded("final_gather", lambda: final_gather(args, index, out, s, rank, world, local_rank, backend, comm_device, by_parts, strong, n_paths, all_steps,
                                                                      my_paths, barrier, dist, torch))
        if not (args.no_extras or args.no_config4):
            if abandoned:
                c4_sharded = {"error": "skipped: " + abandoned[0] + " was abandoned on this rank", "rank": rank}
            else:
                c4_sharded = bounded("config4_sharded", lambda: config4_sharded(args, rank, world, local_rank, backend, comm_device, barrier, dist, torch))
    else:
        all_steps = float(steps_done)

    if rank == 0:
        # cpu_baseline is reported at N = 1 only (rank 0's host cores)
        cpu, parity_checked = None, 0
        if not (args.no_cpu_baseline or world > 1):
            # ... and every run of the baseline is a parity run at headline size: the oracle's walk of the sampled paths against the rows of
            # the last timed extraction (length, sum of node ids, order-dependent hash of every path: gbwt_hip_path_sums / _hashes)
            cpu, o_paths, o_lens, o_sums, o_hashes = cpu_baseline(index_path, n_paths, args.cpu_seconds)
            assert np.array_equal(my_paths[o_paths], o_paths), "paths are not in id order"
            assert np.array_equal(row_lens[o_paths], o_lens), "row lengths differ from the oracle's"
            assert np.array_equal(sums[o_paths], o_sums), "node sums of the extracted paths differ from the oracle's"
            assert np.array_equal(hashes[o_paths], o_hashes), "order-dependent hashes of the extracted paths differ from the oracle's"
            parity_checked = int(len(o_paths))
            cpu["parity_checked_paths"] = parity_checked
        b_per_step, sampled_steps = algorithmic_bytes(index_path, n_paths, args.bytes_sample)
        walk_avg_ms = float(np.mean(walk_ms))
        extras = {}
        if world == 1 and not args.no_extras:
            # (a) the same passes WITHOUT sequence samples: one walker per end of every row, the "one lane per active path" shape
            saved = {k: os.environ.get(k) for k in ("GBWT_HIP_SAMPLE_INTERVAL", "GBWT_HIP_VMM")}
            os.environ["GBWT_HIP_SAMPLE_INTERVAL"] = "0"
            os.environ["GBWT_HIP_VMM"] = "0"      # four passes only: rebuilding the rows from spread chunks (at the third request) would be all they time
            try:
                plain = G.GBZ.load(index_path, device=local_rank)
                _, u_walk, _ = timed_passes(plain, ids, 1)
                t0 = time.perf_counter()
                u_out, u_walk, _ = timed_passes(plain, ids, 3)
                torch.cuda.synchronize()
                u_elapsed = time.perf_counter() - t0
                assert int(u_out.total) == steps_done and np.array_equal(plain.path_sums(len(ids)), truth)
                extras["value_unsampled"] = steps_done * 3 / u_elapsed
                extras["unsampled"] = {"kernel_ms": float(np.mean(u_walk)), "open_ms": plain.open_times()["total_ms"],
                                       "note": "GBWT_HIP_SAMPLE_INTERVAL=0: no sequence samples, every row walked by one lane from each of its two ends"}
                plain.close()
            finally:
                for k, v in saved.items():       # a run launched with one of the knobs set keeps it for what follows (and for the fingerprint)
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            # (b)-(e) the other BASELINE configs, each with a roofline object of its own (tools/configs.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import configs as K
            emitted = "emitted bytes: the u32 node id every LF-step writes (4 B per step) / kernel time; index reads show up in `traffic`"
            # every config with its own bounded CPU leg (5 s of oracle each, compared with what the GPU returned) unless --no-cpu-baseline
            leg_x = None if args.no_cpu_baseline else cpu_leg_extraction(3.0)    # (timed plain walk + the untimed parity walk of the same paths)
            leg_s = None if args.no_cpu_baseline else cpu_leg_search(4.0)
            extras["secondary"] = config_roofline(K.secondary(args.sites, args.haplotypes, model, args.seed, device=local_rank, cpu_leg=leg_x), "secondary", emitted)
            extras["high_degree"] = config_roofline(K.high_degree(args.haplotypes, args.seed, device=local_rank, cpu_leg=leg_x), "high_degree", emitted)
            if not args.no_search:
                extras["search"] = config_roofline(K.search(device=local_rank, cpu_leg=leg_s), "search",
                                                   "bytes a query must move in this layout: its nodes in, its state out, and per step one 64-byte record descriptor + "
                                                   "two 16-byte rank blocks / kernel time (find + 9 x extend, unidirectional)", kernel="k_search")
            if not args.no_config4:
                c4_definition = ("bytes moved by walk + format: node ids written by the walk and read by the formatter + the text written "
                                 "/ wall time of the two requests (P-lines, W-lines), host side included")
                # Config 4 runs in a PROCESS OF ITS OWN (a child: tools/configs.py; this process keeps its GPU state and waits).  Its
                # `first_request_ms` -- the one request gbunzip's flow makes -- includes the allocation of 73 GB of rows and text, and memory a
                # process has given back is paid for by that process's next large allocation (profiles/r05_alloc_microbench.txt: 2.5 s after
                # a release of 48 GiB): behind the configs above, in this process, the first request once took 1.8 s instead of 36 ms.
                # (Round 6 also ran the generator next to this process's configs -- 107 s instead of 140 for the whole line -- and saw first
                # requests of 1.8-2.4 s in those runs; the same then showed up in this order as well: it is the driver clearing released
                # memory under a 73 GB allocation, by the state other processes left the device in, not the order of the processes
                # (tools/vram_first_touch_probe.py).  config4.first_request_device_ms holds it: the clearing runs on the device, in front of the first kernel that touches the allocation.)
                import subprocess
                names = ["config4"] + (["config4_small"] if args.c4_size == "full" else [])
                cmd = [sys.executable, os.path.join(ROOT, "tools", "configs.py")] + names + ["--device", str(local_rank), "--c4-size", args.c4_size]
                if not args.no_cpu_baseline:
                    cmd += ["--cpu-leg", "4"]
                child = subprocess.run(cmd, capture_output=True, text=True)
                sys.stderr.write(child.stderr[-4000:])
                got = None
                for text in reversed(child.stdout.splitlines()):
                    if text.startswith("{"):
                        got = json.loads(text)
                        break
                if child.returncode != 0 or got is None:
                    extras["config4"] = {"error": f"tools/configs.py {' '.join(names)} exited with {child.returncode}: {child.stderr[-400:]}"}
                else:
                    if len(names) == 1:
                        got = {"config4": got}
                    for name in names:
                        extras[name] = config_roofline(got[name], name, c4_definition)
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same command
        # (tools/measure_round.sh -> profiles/*_hbm_traffic.json).  It is quoted only when those passes ran THIS build with
        # THESE knobs on THIS workload (fingerprint of the kernel sources + GBWT_HIP_* environment); a profile of another
        # build says nothing about this run.
        fingerprint = source_fingerprint()
        workload_key = f"sites={args.sites} haplotypes={args.haplotypes} model={args.model} seed={args.seed}"
        traffic, traffic_source = lookup_traffic(workload_key) if world == 1 else (None, "N > 1: no profile")
        seconds = walk_avg_ms * 1e-3
        # What this kernel must move per LF-step whatever happens in the caches: the emitted u32 node id.  (The index it
        # reads -- two-step blocks and descriptors -- is shared by the 64+ steps of a block and mostly survives in L2 /
        # MALL; those bytes are in `traffic`.)  SURVEY 8d's H + P + 4 is the byte count of the REFERENCE's scan of every
        # record up to the offset; rank blocks answer a step without that scan (the run-length decode happens once, at open:
        # open.upload_ms), so that figure is reported next to it as `survey_8d` and not as the fraction of the roofline.
        out_bytes = 4.0 * steps_done
        achieved = out_bytes / seconds / 1e9
        reference_rate = b_per_step * steps_done / seconds / 1e9
        result = {
            "metric": "LF-steps/sec (batched path extract)",
            "value": all_steps * args.steps / elapsed,
            "unit": "LF-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32",   # device arithmetic: record indices, offsets and node ids are 32-bit (the C ABI widens to u64 where the reference has usize)
            "data": "synthetic",
            # paths of the last timed extraction whose length, node sum and order-dependent hash equal the CPU oracle's walk (the cpu_baseline leg)
            "parity_checked_paths": parity_checked,
            # the one-shot flow (load, extract once) next to the steady state `value` is quoted on
            # LF-steps of all ranks / (open + first pass) of the slowest rank; value_cold_incl_init also counts what starting the HIP runtime cost
            "value_cold": all_steps / (cold_ms * 1e-3),
            "value_cold_incl_init": all_steps / ((cold_ms + runtime_init_ms) * 1e-3),
            "kernel_ms_per_rank": rank_kernel_ms,
            # N > 1: one pass + the gather of its rows on rank 0 (the only collective of the job, outside `value`'s timed region)
            "value_incl_gather": None if not (gather_info and "ms" in gather_info) else all_steps / ((elapsed / args.steps) + gather_info["ms"] * 1e-3),
            "open_ms": open_ms,
            "sample_walk_ms": open_times["sample_ms"],
            "first_pass_ms": first_pass_ms,
            "config": {
                "workload": (f"bubble-chain GBZ, {args.haplotypes} paths x {3 * args.sites} nodes ({args.sites} sites, {args.model}, seed {args.seed}), "
                             f"all forward sequences -> device CSR; index replicated, " +
                             (f"every path walked by every GPU over its {world}-th of the way" if by_parts else f"path p walked by GPU p mod {world}")) if strong else
                            (f"bubble-chain GBZ, {args.haplotypes} paths x {3 * args.sites} nodes per GPU "
                             f"({args.sites} sites, {args.model}, seed {args.seed}+rank), all forward sequences -> device CSR"),
                "paths_per_gpu": int(len(ids)),
                "lf_steps_per_gpu": steps_done,
                "index_bytes": int(index.stats.data_bytes),
                "records": int(index.stats.records),
                "sharding": ("one index replicated on every rank, every row cut at sequence samples into G stretches, stretch r of EVERY path on rank r "
                             "(gbwt_hip_extract_part_device), no data-path collective; whole CSR joined on rank 0 afterwards" if by_parts else
                             "one index replicated on every rank, path id p -> rank p mod G, no data-path collective; whole CSR gathered on rank 0 afterwards") if strong
                            else "one contig (index + path set) per rank, no data-path collective",
                "generator_seconds": round(gen_s, 1),
            },
            "open": {
                "open_ms": open_ms, "flags": "GBWT_HIP_OPEN_EXTRACT" if open_flags == G.OPEN_EXTRACT else "GBWT_HIP_OPEN_ALL",
                "parse_ms": open_times["parse_ms"], "upload_ms": open_times["upload_ms"], "sample_ms": open_times["sample_ms"],
                "index_device_bytes": int(index.memory_usage()["index_device_bytes"]),
                "samples": int(open_times["samples"]), "checkpoint_sampling": bool(open_times["checkpoint_sampling"]),
                "checkpoint_walkers": int(open_times["checkpoint_walkers"]), "checkpoint_orphans": int(open_times["checkpoint_orphans"]),
                "first_pass_ms": first_pass_ms, "first_pass_kernel_ms": first_walk_ms, "runtime_init_ms": runtime_init_ms,
                "note": "open_ms = GBZ.load of the .gbz (file read + parse, upload, rank blocks and descriptors, sequence samples) with the HIP "
                        "runtime already started (runtime_init_ms: what starting it cost, once per process -- context, code objects, a tiny open, and "
                        "one 64 MB pageable copy each way, which sets up the runtime's staging for large copies); first_pass_ms includes the "
                        "allocation of the rows; value_cold = LF-steps / (open_ms + first_pass_ms)",
            },
            "roofline": {
                "bound": "hbm",
                "definition": "emitted bytes: the u32 node id every LF-step writes (4 B per step) / kernel time -- a STORE roofline; SURVEY 8(d)'s per-step "
                              "figure (what the reference's scan reads) is priced in `cold` (with the open, where the run-length decode happens) and in `survey_8d`",
                "kernel": "k_walk_direct",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_step": 4.0,
                "algorithmic_note": "bytes this kernel must move per LF-step: the emitted u32 node id, written once "
                                    "(index reads are shared by all steps of a rank block and show up in `traffic`)",
                "traffic": traffic,
                "traffic_frac": None if traffic is None else traffic / seconds / 1e9 / HBM_PEAK_GBS,
                "traffic_source": traffic_source,
                "survey_8d": {
                    "bytes_per_step": b_per_step,
                    "achieved": reference_rate,
                    "frac": reference_rate / HBM_PEAK_GBS,
                    "note": "SURVEY 8d's definition: H + P + 4 = record header + run stream scanned up to the offset + emitted id per step, i.e. what "
                            "an implementation with the reference's access pattern would read to keep this pace.  Above the peak because the "
                            "run-length decode is done once, at open (rank blocks, open.upload_ms), and the timed step is a popcount on a "
                            "16-byte packed half-block -- an algorithmic gain, not a fraction of the roofline",
                    "sample": f"exact over {sampled_steps} LF-steps of {min(args.bytes_sample, n_paths)} paths, scaled to {steps_done}",
                },
                "cold": {
                    "bytes": b_per_step * all_steps, "ms": cold_ms, "achieved": b_per_step * all_steps / (cold_ms * 1e-3) / 1e9,
                    "frac": b_per_step * all_steps / (cold_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
                    "note": "SURVEY 8(d)'s algorithmic bytes (H + P + 4 per step) / (open_ms + first_pass_ms) / peak: the one figure that contains ALL the work "
                            "8(d) prices -- the run-length decode into rank blocks and the sampling walk at open, then one extraction",
                },
                "kernel_ms": walk_avg_ms,
                "extract_ms": float(np.mean(total_ms)),
                "source_fingerprint": fingerprint,
            },
        }
        result.update(extras)
        if world > 1:
            result["shard"] = args.shard if strong else None
            result["other_cut"] = other_cut          # the same batch under the other partition, same K passes: both on every N > 1 line
            if c4_sharded is not None:
                result["config4"] = c4_sharded       # BASELINE config 4 as named: sharded over the N GPUs, RCCL gather of the text
        if cpu is not None:
            result["cpu_baseline"] = cpu
        if gather_info is not None:
            result["config"]["final_gather"] = gather_info
        if abandoned:
            result["abandoned"] = abandoned
        if failed_checks:
            result["errors"] = [f"{w}: the step's result failed its check (see its object)" for w in failed_checks]
        result["bench_seconds"] = round(time.perf_counter() - _PROCESS_START, 1)   # this process, imports to line (generators, CPU legs and the other configs included)
        print(json.dumps(result), flush=True)
    if abandoned:
        # a thread of this process still sits in a collective that will not complete: no barrier, no teardown of communicators it holds --
        # the line is out (rank 0), the files go, the process ends here
        for f in [index_path] + [x + tail for x in _C4_FILES for tail in ("", ".generic.npy", ".tmp", ".lines.gfa")]:
            try:
                os.remove(f)
            except OSError:
                pass
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001  (a rank that failed above: the line is out, nothing left to agree on)
            pass
    try:
        os.remove(index_path)
        os.rmdir(tmpdir)
    except OSError:
        pass


if __name__ == "__main__":
    main()
