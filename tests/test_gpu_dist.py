"""The RCCL exchange between real ranks: one process per GPU of the box, backend "nccl" (= RCCL), world size = the number of GPUs.
Skipped on a one-GPU box, where tests/test_gpu_gfa.py::test_device_resident_lines_and_rccl_gather runs the same plumbing with world
size 1 and tests/test_dist_cpu.py the N > 1 logic over gloo."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the TEST build of the library (csrc/Makefile): the product objects + comm.hip compiled with -DGBWT_HIP_TEST_TRANSPORT, i.e. with the
# loopback transport (ranks = threads of one process on one GPU) and the self-send switch.  libgbwt_hip.so itself holds neither.
TEST_LIB = os.path.join(ROOT, "gbwt_rs_amd", "csrc", "libgbwt_hip_testtransport.so")

CHILD = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
import gbwt_rs_amd as G
from gbwt_rs_amd import dist as D, synth as S
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
device = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=device)
# the same index on every rank (replicated, SURVEY 8e), paths dealt p -> rank p mod world
s = S.Synth.chain(sites=SITES, haplotypes=HAPLOTYPES, alleles=2, model=S.MOSAIC, founders=16, switch_rate=5e-3, seed=5, extra=1, indel_every=7)
dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, device=local)
mine = np.arange(rank, s.paths, world, dtype=np.uint64)
for attempt in range(4):                      # from the third request on the rows of a large batch are mapped from spread chunks (virtual-memory API)
    out = dev.extract_device(2 * mine)
offsets, nodes = D.paths_tensors(out, device)
lengths = (offsets[1:] - offsets[:-1]).clone()
# straight from the workspace's rows (no staging copy): RCCL point-to-point sends out of whatever memory the rows live in
g_off, g_val = D.gather_rows(lengths, nodes, dst=0, interleaved=True)
if rank == 0:
    assert g_off.numel() == s.paths + 1
    g_off = g_off.cpu().numpy()
    for p in list(range(0, s.paths, max(1, s.paths // 37))) + [s.paths - 1]:
        row = g_val[g_off[p]:g_off[p + 1]].cpu().numpy().astype(np.uint32)
        assert np.array_equal(row, s.path(p)), p
    assert int(g_off[-1]) == (s.size - s.sequences) // 2
len_parts, val_parts = D.gather_parts(lengths, nodes, dst=0)
if rank == 0:
    assert sum(int(v.numel()) for v in val_parts) == (s.size - s.sequences) // 2
    order = list(D.rows_in_path_order(len_parts))
    assert len(order) == s.paths and order[1][0] == 1 % world
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("RCCL_RANKS_OK", world, int(out.total) * 4)
'''


# The same exchange through the C ABI (gbwt_hip_comm_*: RCCL called by libgbwt_hip.so itself, rows placed by kernels): rows and GFA lines,
# interleaved and contiguous shards, against the generator and against one rank formatting everything alone.
CHILD_CAPI = r'''
import os, sys, hashlib
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
import gbwt_rs_amd as G
from gbwt_rs_amd import dist as D, synth as S
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
device = torch.device("cuda", local)
dist.init_process_group("nccl" if world > 1 else "gloo", **({"device_id": device} if world > 1 else {}))
comm = D.Comm(rank, world, local)
path = os.path.join(TMP, "capi_ranks.gbz")
if rank == 0:
    g = S.Synth.genome(contigs=6, fragments=3, haplotypes=30, sites=SITES, seed=9)
    g.save(path + ".tmp", as_gbz=True)
    os.replace(path + ".tmp", path)
dist.barrier()
gbz = G.GBZ.load(path, device=local)
n_paths = gbz.stats.paths
everything = np.arange(n_paths, dtype=np.uint64)
for interleaved in (True, False):
    mine = D.shard_ids(everything, rank, world, interleaved=interleaved)
    for attempt in range(ATTEMPTS):           # from the third request on large rows are mapped from spread chunks (GBWT_HIP_VMM)
        out = gbz.extract_device(2 * mine)
    got = comm.gather_rows(gbz, root=0, interleaved=interleaved)
    stats = comm.last()
    if rank == 0:
        off, nodes = D.paths_tensors(got, device)
        off, nodes = off.cpu().numpy(), nodes.cpu().numpy().astype(np.uint32)
        w_off, w_nodes = gbz.sequences_csr(2 * everything)
        assert np.array_equal(off, w_off.astype(np.int64)) and np.array_equal(nodes, w_nodes), interleaved
    else:
        assert got is None and stats["bytes"] == 4 * int(out.total) + 8 * len(mine)
    lines = gbz.path_lines_device(mine, 1)
    got = comm.gather_lines(gbz, root=0, interleaved=interleaved)
    if rank == 0:
        off, text = D.lines_tensors(got, device)
        alone = gbz.path_lines(everything, 1)
        assert bytes(text.cpu().numpy().tobytes()) == alone and int(off[-1]) == len(alone) and off.numel() == n_paths + 1, interleaved
# every rank its stretch of EVERY row (gbwt_hip_extract_part_device), joined on rank 0 (GBWT_HIP_GATHER_PARTS)
for attempt in range(ATTEMPTS):
    out = gbz.extract_part_device(2 * everything, rank, world)
got = comm.gather_rows(gbz, root=0, layout=D.GATHER_PARTS)
if rank == 0:
    off, nodes = D.paths_tensors(got, device)
    off, nodes = off.cpu().numpy(), nodes.cpu().numpy().astype(np.uint32)
    w_off, w_nodes = gbz.sequences_csr(2 * everything)
    assert int(got.n) == n_paths and np.array_equal(off, w_off.astype(np.int64)) and np.array_equal(nodes, w_nodes), "parts of rows"
else:
    assert got is None
dist.barrier()
comm.close()
dist.destroy_process_group()
if rank == world - 1:
    print("PEER_STATS", stats, flush=True)      # (the last rank is a peer whenever there is one: did its rows go through a staging copy?)
if rank == 0:
    print("CAPI_RANKS_OK", world, int(out.total) * 4, flush=True)
'''


CHILD_LOOPBACK = r'''
import os, sys, threading, faulthandler
faulthandler.dump_traceback_later(90, exit=True)     # a rank that waits for ever: every thread's stack on stderr, and out
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gbwt_rs_amd as G
from gbwt_rs_amd import dist as D, synth as S
device = torch.device("cuda", 0)
path = os.path.join(TMP, "loopback.gbz")
S.Synth.genome(contigs=5, fragments=3, haplotypes=22, sites=SITES, seed=11).save(path, as_gbz=True)
alone = G.GBZ.load(path, device=0)
n_paths = alone.stats.paths
everything = np.arange(n_paths, dtype=np.uint64)
w_off, w_nodes = alone.sequences_csr(2 * everything)
w_lines = alone.path_lines(everything, 1)
failures = []

def rank_main(rank, world, box, ready):
    try:
        def broadcast(raw):
            if raw is not None:
                box.append(raw); ready.set()
            ready.wait()
            return box[0]
        gbz = G.GBZ.load(path, device=0)
        comm = D.Comm(rank, world, 0, broadcast=broadcast)
        for root in (0, world - 1):
            for layout in (D.GATHER_BLOCKS, D.GATHER_INTERLEAVED, D.GATHER_PARTS):
                if layout == D.GATHER_PARTS:
                    out = gbz.extract_part_device(2 * everything, rank, world)
                else:
                    mine = D.shard_ids(everything, rank, world, interleaved=layout == D.GATHER_INTERLEAVED)
                    out = gbz.extract_device(2 * mine)
                got = comm.gather_rows(gbz, root=root, layout=layout)
                if rank == root:
                    off, nodes = D.paths_tensors(got, device)
                    assert int(got.n) == n_paths and int(got.total) == len(w_nodes), (layout, int(got.n), int(got.total))
                    assert np.array_equal(off.cpu().numpy(), w_off.astype(np.int64)) and np.array_equal(nodes.cpu().numpy().astype(np.uint32), w_nodes), ("rows", world, root, layout)
                else:
                    assert got is None
                if layout != D.GATHER_PARTS:
                    gbz.path_lines_device(mine, 1)
                    got = comm.gather_lines(gbz, root=root, interleaved=layout == D.GATHER_INTERLEAVED)
                    if rank == root:
                        off, text = D.lines_tensors(got, device)
                        assert bytes(text.cpu().numpy().tobytes()) == w_lines and off.numel() == n_paths + 1, ("lines", world, root, layout)
        comm.close()
    except BaseException as e:
        import traceback
        failures.append((rank, world, traceback.format_exc()))
        print("RANK FAILED", rank, world, traceback.format_exc(), file=sys.stderr, flush=True)
        ready.set()
        os._exit(3)                                   # the other ranks wait for this one inside a collective: nothing to join

for world in WORLDS:
    box, ready = [], threading.Event()
    threads = [threading.Thread(target=rank_main, args=(r, world, box, ready), daemon=True) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join(60)
    if any(t.is_alive() for t in threads):
        faulthandler.dump_traceback(all_threads=True)
        print("A RANK HANGS", world, file=sys.stderr, flush=True)
        os._exit(4)
    assert not failures, failures[0][2]
print("LOOPBACK_OK", WORLDS, n_paths, len(w_nodes), flush=True)
'''


def test_capi_gather_between_loopback_ranks(tmp_path):
    """Everything of comm.hip except RCCL itself, for world sizes 2, 3 and 8 on ONE GPU: the ranks are threads of one process
    (libgbwt_hip_testtransport.so with GBWT_HIP_COMM_LOOPBACK=1: the all-gather and the point-to-point group served by device-to-device copies with RCCL's matching rules),
    each with its own index handle, workspace and communicator.  Blocks of rows, interleaved rows and stretches of every row
    (gbwt_hip_extract_part_device) gathered on the first and on the last rank, GFA lines too; against one handle extracting alone."""
    script = f"ROOT = {ROOT!r}; SITES = 400; TMP = {str(tmp_path)!r}; WORLDS = (2, 3, 8)\n" + CHILD_LOOPBACK
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=dict(os.environ, GBWT_HIP_COMM_LOOPBACK="1", GBWT_HIP_LIB=TEST_LIB))
    if out.returncode != 0 and os.path.isdir(os.path.join(ROOT, "gpurun_out")):      # every thread's stack, where a gpurun call can take it home
        open(os.path.join(ROOT, "gpurun_out", "loopback_failure.txt"), "w").write(out.stdout + "\n" + out.stderr)
    assert out.returncode == 0 and "LOOPBACK_OK" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-6000:])


CHILD_C4_LOOPBACK = r'''
import os, sys, threading, faulthandler, json
faulthandler.dump_traceback_later(400, exit=True)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import gbwt_rs_amd as G
from gbwt_rs_amd import dist as D
import c4_bench
path = os.path.join(TMP, "c4.gbz")
g = c4_bench.generate(SIZE, path)
generic = np.load(path + ".generic.npy")
walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
results = {}
shared = [None] * 8

def rank_main(rank, world, box, ready, barrier):
    try:
        def broadcast(raw):
            if raw is not None:
                box.append(raw); ready.set()
            ready.wait()
            return box[0]
        gbz = G.GBZ.load(path, device=0)                     # every rank its own replica of the index, as one process per GPU has
        comm = D.Comm(rank, world, 0, broadcast=broadcast)
        def allgather(value):                                # (between threads: a shared list, two barriers)
            shared[rank] = value
            barrier.wait()
            got = list(shared[:world])
            barrier.wait()
            return got
        res, my = c4_bench.run_sharded(gbz, generic, walks, rank, world, comm, barrier.wait, 0, passes=2, file_path=path + f".{world}.gfa", allgather=allgather)
        results[(world, rank)] = (res, my)
        comm.close(); gbz.close()
    except BaseException:
        import traceback
        print("RANK FAILED", rank, world, traceback.format_exc(), file=sys.stderr, flush=True)
        os._exit(3)

for world in WORLDS:
    box, ready, barrier = [], threading.Event(), threading.Barrier(world)
    threads = [threading.Thread(target=rank_main, args=(r, world, box, ready, barrier), daemon=True) for r in range(world)]
    for t in threads: t.start()
    for t in threads: t.join(300)
    if any(t.is_alive() for t in threads):
        faulthandler.dump_traceback(all_threads=True)
        os._exit(4)
    root = results[(world, 0)][0]
    assert root["sharded_file"]["bytes"] == root["text_bytes"] + root["p_text_bytes"]
    assert root["check"] and root["text_bytes"] == sum(results[(world, r)][1]["text_bytes"] for r in range(world)) - root["p_text_bytes"], root
    assert sum(results[(world, r)][1]["walks"] for r in range(world)) == len(walks)
print("C4_LOOPBACK_OK", WORLDS, len(walks), json.dumps(results[(WORLDS[-1], 0)][0]), flush=True)
'''


def test_config4_sharded_between_loopback_ranks(tmp_path):
    """BASELINE config 4's N > 1 flow exactly as bench.py --gpus N runs it (tools/c4_bench.py: run_sharded) -- rank r walks and formats ITS
    block of path ids, gbwt_hip_gather_lines puts W- and P-lines in path order on rank 0, every gathered byte compared on the device with
    rank 0 formatting alone -- for world sizes 8 and 3 on ONE GPU, the ranks being threads of one process over the loopback transport of
    the test build (each with its own index replica, workspace and communicator).  What is left for real hardware is RCCL itself."""
    script = f"ROOT = {ROOT!r}; SIZE = 'medium'; TMP = {str(tmp_path)!r}; WORLDS = (8, 3)\n" + CHILD_C4_LOOPBACK
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=900, env=dict(os.environ, GBWT_HIP_COMM_LOOPBACK="1", GBWT_HIP_LIB=TEST_LIB))
    assert out.returncode == 0 and "C4_LOOPBACK_OK" in out.stdout, (out.returncode, out.stdout[-2000:], out.stderr[-6000:])


def run_capi_ranks(world, sites, attempts, tmp, timeout, extra_env=None):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = f"ROOT = {ROOT!r}; SITES = {sites}; ATTEMPTS = {attempts}; TMP = {str(tmp)!r}\n" + CHILD_CAPI
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "--no-python", sys.executable, "-c", script]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("self_send", [False, True])
def test_capi_comm_on_one_gpu(tmp_path, self_send):
    """gbwt_hip_comm_* with world size 1 on the one GPU of a box: RCCL loaded by the library, a communicator, the all-gather of the
    counts, the placement kernels (interleaved and contiguous, node ids and ragged GFA lines at every byte alignment) -- and, with
    GBWT_HIP_COMM_SELF_SEND, the root's own part through ncclSend / ncclRecv in one group.  The peers are what is missing."""
    out = run_capi_ranks(1, 300, 1, tmp_path, 600, {"GBWT_HIP_COMM_SELF_SEND": "1", "GBWT_HIP_LIB": TEST_LIB} if self_send else None)
    assert out.returncode == 0 and "CAPI_RANKS_OK" in out.stdout, out.stderr[-3000:]


def test_capi_comm_between_ranks(tmp_path):
    """The same with one rank per GPU: the point-to-point group over xGMI, every rank's rows and lines in path order on rank 0."""
    world = gpus()
    if world < 2:
        pytest.skip("one GPU on this box: ranks need devices of their own for RCCL point-to-point")
    out = run_capi_ranks(world, 3000, 4, tmp_path, 900)
    assert out.returncode == 0 and "CAPI_RANKS_OK" in out.stdout, out.stderr[-3000:]


@pytest.mark.parametrize("direct", ["0", "1"])
def test_capi_comm_sends_from_mapped_rows(tmp_path, direct):
    """Rows that the workspace has rebuilt from chunks of the virtual-memory API (GBWT_HIP_VMM with a 64 MiB threshold, so that a test-size
    batch is mapped): staged through an ordinary allocation before the send (default), or sent straight out of the mapping
    (GBWT_HIP_COMM_DIRECT=1) -- the mode a multi-GPU box has to prove before it becomes the default."""
    world = gpus()
    if world < 2:
        pytest.skip("one GPU on this box: ranks need devices of their own for RCCL point-to-point")
    out = run_capi_ranks(world, 40000, 4, tmp_path, 1800, {"GBWT_HIP_VMM": "64:2:64:2", "GBWT_HIP_COMM_DIRECT": direct})
    assert out.returncode == 0 and "CAPI_RANKS_OK" in out.stdout, out.stderr[-3000:]
    peer = [l for l in out.stdout.splitlines() if l.startswith("PEER_STATS")][-1]
    assert ("'staged_send': True" in peer) == (direct == "0"), peer


def run_ranks(world, sites, haplotypes, timeout):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = f"ROOT = {ROOT!r}; SITES = {sites}; HAPLOTYPES = {haplotypes}\n" + CHILD
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "--no-python", sys.executable, "-c", script]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


def gpus():
    import torch
    return torch.cuda.device_count()        # counting devices does not start the runtime in this process


def test_rccl_gather_between_ranks():
    """dist.gather_rows / gather_parts over RCCL with one rank per GPU: interleaved shards of one path set, rows back in path order on
    rank 0, every 37th row against the generator."""
    world = gpus()
    if world < 2:
        pytest.skip("one GPU on this box: ranks need devices of their own for RCCL point-to-point")
    out = run_ranks(world, 3000, 700, 900)
    assert out.returncode == 0 and "RCCL_RANKS_OK" in out.stdout, out.stderr[-3000:]


def test_rccl_send_from_a_mapped_rows_buffer():
    """The same with a batch whose rows exceed 4 GiB per rank, i.e. rows that the workspace has rebuilt from 2 GiB chunks of the
    virtual-memory API (capi_internal.hpp: DeviceBuffer): RCCL sends straight out of that mapping."""
    world = gpus()
    if world < 2:
        pytest.skip("one GPU on this box: ranks need devices of their own for RCCL point-to-point")
    out = run_ranks(world, 110000 * world, 5000, 3000)   # 5 000 / world paths of 236 000 x world nodes each: 4.7 GB of node ids per rank
    assert out.returncode == 0 and "RCCL_RANKS_OK" in out.stdout, out.stderr[-3000:]
    assert int(out.stdout.split("RCCL_RANKS_OK")[1].split()[1]) >= 4 << 30
