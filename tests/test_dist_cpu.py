"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing: sharding of the path set and the final ordered gather.
The per-rank "extraction" is done by the oracle here (test infrastructure); on the GPU box the same code moves
device tensors over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from gbwt_rs_amd import dist as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, interleaved, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle = O.OracleGBWT.load(os.path.join(O.GOLDEN, "with-empty.gbwt"))
        ids = np.arange(oracle.sequences(), dtype=np.uint64)
        mine = D.shard_ids(ids, rank, world, interleaved=interleaved)
        offsets, nodes = oracle.extract(mine)
        lengths = torch.from_numpy(np.diff(offsets).astype(np.int64))
        values = torch.from_numpy(nodes.astype(np.int64))
        g_off, g_val = D.gather_rows(lengths, values, dst=0, interleaved=interleaved)
        if rank == 0:
            full_off, full_nodes = oracle.extract(ids)
            assert np.array_equal(g_off.numpy(), full_off.astype(np.int64))
            assert np.array_equal(g_val.numpy(), full_nodes.astype(np.int64))
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        else:
            assert g_off is None and g_val is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("interleaved", [False, True])
def test_shard_and_gather_world2(tmp_path, interleaved):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, interleaved, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _parts_worker(rank, world, port, out_dir):
    """Parts of rows (bench.py --shard parts, gbwt_hip_extract_part_device): every rank holds ITS stretch of every row -- cut here at
    rank-dependent points of the oracle's rows, empty stretches and empty rows included --, gather_parts moves them and join_row_parts on
    rank 0 must give back the rows."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle = O.OracleGBWT.load(os.path.join(O.GOLDEN, "with-empty.gbwt"))
        ids = np.arange(oracle.sequences(), dtype=np.uint64)
        full_off, full_nodes = oracle.extract(ids)
        lens = np.diff(full_off).astype(np.int64)
        cut = lambda r: np.minimum(lens, (lens * r + (r * 3) % world) // world)      # cuts that differ from row to row; cut(0) = 0, cut(world) = len
        lo, hi = cut(rank), (lens if rank + 1 == world else cut(rank + 1))
        mine = np.concatenate([full_nodes[int(full_off[k] + lo[k]):int(full_off[k] + hi[k])] for k in range(len(ids))] + [np.zeros(0, dtype=full_nodes.dtype)])
        len_parts, val_parts = D.gather_parts(torch.from_numpy(hi - lo), torch.from_numpy(mine.astype(np.int64)), dst=0)
        if rank == 0:
            j_off, j_val = D.join_row_parts(len_parts, val_parts)
            assert np.array_equal(j_off.numpy(), full_off.astype(np.int64)) and np.array_equal(j_val.numpy(), full_nodes.astype(np.int64))
            open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_parts_of_rows_joined_on_rank0(tmp_path, world):
    port = _free_port()
    mp.spawn(_parts_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def _lines_worker(rank, world, port, interleaved, out_dir):
    """The final GFA concatenation: every rank formats the W-lines of its shard of the paths (oracle here, the device
    formatter on a GPU box) and rank 0 must end up with the bytes of one pass over all paths, in path order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gbz = O.OracleGBZ(os.path.join(O.GOLDEN, "example.gbz"))
        ids = np.arange(6, dtype=np.uint64)
        mine = D.shard_ids(ids, rank, world, interleaved=interleaved)
        lines = [gbz.path_lines([int(p)], 1) for p in mine]
        text = torch.frombuffer(bytearray(b"".join(lines)), dtype=torch.uint8) if lines and sum(map(len, lines)) else torch.empty(0, dtype=torch.uint8)
        offsets = torch.zeros(len(lines) + 1, dtype=torch.int64)
        offsets[1:] = torch.cumsum(torch.tensor([len(x) for x in lines], dtype=torch.int64), 0) if lines else offsets[1:]
        g_off, g_text = D.gather_lines(offsets, text, dst=0, interleaved=interleaved)
        if interleaved:   # the copy-free form a file writer uses: rows in path order where they arrived
            len_parts, text_parts = D.gather_parts(offsets[1:] - offsets[:-1], text, dst=0)
            if rank == 0:
                streamed = b"".join(text_parts[r][a:b].numpy().tobytes() for r, a, b in D.rows_in_path_order(len_parts))
                assert streamed == gbz.path_lines([int(p) for p in ids], 1)
            else:
                assert len_parts is None and text_parts is None
        if rank == 0:
            assert bytes(g_text.numpy().tobytes()) == gbz.path_lines([int(p) for p in ids], 1)
            assert int(g_off[-1]) == g_text.numel()
            if interleaved:
                assert g_off.numel() == 7     # one row per line, in path order
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        else:
            assert g_off is None and g_text is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,interleaved", [(2, False), (2, True), (4, True), (3, False)])
def test_gather_gfa_lines(tmp_path, world, interleaved):
    port = _free_port()
    mp.spawn(_lines_worker, args=(world, port, interleaved, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def _ragged_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        n = 23                                                    # not a multiple of the world size; some rows empty
        lens = rng.integers(0, 9, n)
        lens[[0, 7, 22]] = 0
        rows = [np.arange(l, dtype=np.int64) + 1000 * k for k, l in enumerate(lens)]
        for interleaved in (False, True):
            mine = D.shard_ids(np.arange(n), rank, world, interleaved=interleaved)
            lengths = torch.tensor([len(rows[k]) for k in mine], dtype=torch.int64)
            values = torch.from_numpy(np.concatenate([rows[k] for k in mine]) if len(mine) else np.zeros(0, dtype=np.int64))
            g_off, g_val = D.gather_rows(lengths, values, dst=1, interleaved=interleaved)   # a root that is not rank 0
            if rank == 1:
                assert np.array_equal(np.diff(g_off.numpy()), lens)
                assert np.array_equal(g_val.numpy(), np.concatenate(rows))
            else:
                assert g_off is None
        if rank == 1:
            open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_gather_ragged_rows_world3(tmp_path):
    port = _free_port()
    mp.spawn(_ragged_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 5000, 5001):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    ids = np.arange(10)
    assert sorted(np.concatenate([D.shard_ids(ids, r, 3, interleaved=True) for r in range(3)]).tolist()) == list(range(10))


def test_unique_id_travels_whole():
    """The 128 bytes of a communicator id are binary (an ncclUniqueId): a NUL in the middle must not cut it short on the way to the other
    ranks (round 4: the id was read through a c_char field, i.e. as a C string, and every rank but the first joined a world of its own)."""
    import ctypes as C
    from gbwt_rs_amd import _lib
    raw = bytes([7, 0, 0, 9] + [(3 * k) % 256 for k in range(124)])
    uid = D.unpack_unique_id(raw)
    assert C.sizeof(uid) == 128 and D.pack_unique_id(uid) == raw and bytes(uid.bytes) == raw
    with pytest.raises(ValueError):
        D.unpack_unique_id(raw[:40])


def test_comm_failure_of_rank0_is_collective(monkeypatch):
    """Rank 0 cannot make a communicator id (RCCL not loadable: GBWT_HIP_UNSUPPORTED, the documented fallback case): EVERY rank must
    raise -- rank 0 broadcasts an error marker instead of leaving alone while the peers sit in the broadcast."""
    import threading
    from gbwt_rs_amd import _lib

    real = _lib.lib()

    class NoRccl:
        def __getattr__(self, name):
            return getattr(real, name)

        def gbwt_hip_comm_unique_id(self, out):
            return _lib.UNSUPPORTED

    monkeypatch.setattr(_lib, "lib", lambda: NoRccl())
    world, box, ready, raised = 3, [], threading.Event(), []

    def broadcast(raw):
        if raw is not None:
            box.append(raw)
            ready.set()
        assert ready.wait(30)
        return box[0]

    def rank_main(rank):
        try:
            D.Comm(rank, world, 0, broadcast=broadcast)
        except _lib.GbwtHipError as e:
            raised.append((rank, e.status))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert sorted(raised) == [(r, _lib.UNSUPPORTED) for r in range(world)]


def _c4_worker(rank, world, port, path, out_dir):
    """Config 4's partition over gloo (tools/c4_bench.py: run_sharded, with the oracle as the per-rank formatter): rank r formats the W-lines
    of ITS block of the walks and the P-lines of its block of the generic paths; rank 0 must hold the bytes of one pass over all of them."""
    import hashlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gbz = O.OracleGBZ(path)
        generic = np.load(path + ".generic.npy")
        walks = np.setdiff1d(np.arange(gbz.gbwt().sequences() // 2, dtype=np.uint64), generic)
        for ids, mode in ((walks, 1), (generic, 0)):
            lo, hi = D.shard_bounds(len(ids), rank, world)
            lines = [gbz.path_lines([int(p)], mode) for p in ids[lo:hi]]
            text = torch.frombuffer(bytearray(b"".join(lines)), dtype=torch.uint8) if lines else torch.empty(0, dtype=torch.uint8)
            offsets = torch.zeros(len(lines) + 1, dtype=torch.int64)
            if lines:
                offsets[1:] = torch.cumsum(torch.tensor([len(x) for x in lines], dtype=torch.int64), 0)
            g_off, g_text = D.gather_lines(offsets, text, dst=0)
            if rank == 0:
                alone = gbz.path_lines([int(p) for p in ids], mode)
                assert hashlib.sha256(g_text.numpy().tobytes()).hexdigest() == hashlib.sha256(alone).hexdigest(), mode
        if rank == 0:
            open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_config4_partition_over_gloo(tmp_path, world):
    """BASELINE config 4's N > 1 flow at rehearsal size ("tiny": 6 contigs x 4 components, labels of 1..1024 bp, one contig whose W-line
    ends pass 2^32): walks dealt to ranks in blocks of path ids, lines gathered on rank 0, sha256 == one rank alone.  The generator's
    ground truth of every line (header fields, end coordinate = fragment + summed label lengths, length) is checked against the oracle
    first -- it is what the full-size GPU test trusts."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import c4_bench
    path = str(tmp_path / "c4_tiny.gbz")
    g = c4_bench.generate("tiny", path, threads=2)
    oracle = O.OracleGBZ(path)
    generic = set(g.generic_paths())
    past = 0
    for p in range(g.paths):
        if p in generic:
            continue
        line = oracle.path_lines([p], 1)
        sample, contig, phase, fragment = (int(x) for x in g.path_names[p])
        nodes, digits, bp = g.path_text_stats(p)
        header = f"W\t{g.sample_names[sample]}\t{phase}\tchr{contig + 1}\t{fragment}\t{fragment + bp}\t".encode()
        assert line.startswith(header) and len(line) == len(header) + digits + nodes + 1, p
        past += fragment + bp > 1 << 32
    assert past > 0, "no W-line of the rehearsal ends past 2^32"
    port = _free_port()
    mp.spawn(_c4_worker, args=(world, port, path, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").exists()


def _rehearse_bench(world, extra_env=None, timeout=420):
    """tests/bench_rehearsal.py under torch.distributed.run, `world` gloo ranks on CPU: (exit code, the JSON lines on stdout, stderr)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "tests", "bench_rehearsal.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--sites", "300", "--haplotypes", "24", "--c4-size", "tiny"]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(x) for x in done.stdout.splitlines() if x.startswith("{")]
    return done.returncode, lines, done.stderr


@pytest.mark.parametrize("world", [2, 3])
def test_bench_n_gpus_over_gloo(world):
    """bench.py --gpus N END TO END without a GPU: N gloo ranks run bench.main() unchanged over a stand-in for the handle
    classes that answers every extraction and every GFA line with the oracle (tests/bench_rehearsal.py).  Every collective of the N > 1 flow
    is entered by every rank (the run ends), rank 0 prints exactly ONE line, and the line has the shape the driver and the judge read:
    the headline under both cuts, the final gather, config 4 sharded with per-rank numbers, the gather and the per-rank file writes."""
    rc, lines, err = _rehearse_bench(world)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, (len(lines), err[-2000:])
    line = lines[0]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                "value_cold", "kernel_ms_per_rank", "value_incl_gather", "open", "shard", "other_cut", "config4"):
        assert key in line, key
    assert line["n_gpus"] == world and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong" and line["vs_baseline"] is None
    assert line["unit"] == "LF-steps/s" and line["higher_is_better"] is True and line["dtype"] == "u32" and "cpu_baseline" not in line   # (N = 1 only)
    assert line["shard"] == "parts" and line["other_cut"]["shard"] == "paths" and line["other_cut"]["value"] > 0
    assert len(line["kernel_ms_per_rank"]) == world
    assert line["config"]["lf_steps_per_gpu"] * world >= 24 * 2 * 300            # every rank its stretch of every row: together all LF-steps
    gather = line["config"]["final_gather"]
    assert "error" not in gather and gather["ms"] > 0 and gather["bytes"] == 4 * 24 * 2 * 300 + 8 * 24, gather
    assert line["value_incl_gather"] is not None and 0 < line["value_incl_gather"] < line["value"]
    c4 = line["config4"]
    assert "error" not in c4, c4
    assert c4["n_gpus"] == world and len(c4["ranks"]) == world and sorted(r["rank"] for r in c4["ranks"]) == list(range(world))
    assert sum(r["walks"] for r in c4["ranks"]) + sum(r["p_lines"] for r in c4["ranks"]) == c4["paths"]
    assert c4["gather_ms"] > 0 and c4["value"] > 0 and c4["value_incl_gather"] < c4["value"] and c4["text_bytes"] == sum(r["text_bytes"] for r in c4["ranks"]) - c4["p_text_bytes"]
    assert c4["sharded_file"]["bytes"] == c4["text_bytes"] + c4["p_text_bytes"] and "gathered W-lines == rank 0" in c4["check"]
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["traffic"] is None                     # (no PMC profile at N > 1)


@pytest.mark.parametrize("failure", ["c4_open:1", "c4_open:0", "c4_generate"])
def test_bench_survives_a_rank_that_fails_in_config4(failure):
    """A rank that cannot open config 4's index (or rank 0 whose generator fails) must not leave its peers waiting in the next collective
    until the process group's timeout -- with RCCL the watchdog would abort them and the headline's line would be lost with them (ADVICE
    r05): the ranks AGREE on the failure before anybody enters a collective (bench.py: config4_sharded.agree), all of them leave the
    object, and rank 0 still prints the line, with the error in `config4`, within seconds."""
    import time
    t0 = time.perf_counter()
    rc, lines, err = _rehearse_bench(2, {"REHEARSAL_FAIL": failure}, timeout=240)
    assert rc == 0 and len(lines) == 1, err[-3000:]
    assert time.perf_counter() - t0 < 200
    line = lines[0]
    assert "error" in line["config4"] and line["value"] > 0 and line["other_cut"]["value"] > 0 and "ms" in line["config"]["final_gather"]


def test_bench_abandons_a_step_that_hangs():
    """A transport that HANGS (not fails) in the final gather -- rank 1 never comes back from the exchange, rank 0 waits for its rows: what a first
    run between GPUs could look like.  Every rank's guard abandons the step after BENCH_GUARD_SECONDS, config 4 behind it is skipped, rank 0
    prints the line with everything measured before (the headline under both cuts), the ranks exit 0 without another barrier, and the whole
    run is bounded."""
    import time
    t0 = time.perf_counter()
    rc, lines, err = _rehearse_bench(2, {"REHEARSAL_FAIL": "gather_hang:1", "BENCH_GUARD_SECONDS": "6"}, timeout=240)
    assert rc == 0 and len(lines) == 1, err[-3000:]
    assert time.perf_counter() - t0 < 120
    line = lines[0]
    assert line["abandoned"] == ["final_gather"] and "abandoned" in line["config"]["final_gather"]["error"] and "skipped" in line["config4"]["error"]
    assert line["value"] > 0 and line["other_cut"]["value"] > 0 and line["n_gpus"] == 2 and line["value_incl_gather"] is None
