"""Multi-GPU plumbing for path extraction: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path set shards by sequence: every rank walks its own shard against its own replica (or its own contig) of the
index, and there is NO collective inside the walk.  The only exchange is the one the reference's writer mutex
stands for (src/bin/gbunzip.rs:421-434): putting the extracted rows -- node ids or finished GFA lines -- back into
path order on one rank.  That is

    1. all_gather of the per-rank row and value counts                (16 B per rank)
    2. variable-size gather of lengths and values to the destination  (ONE group of point-to-point operations:
       dist.batch_isend_irecv, i.e. ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd on RCCL, so that all
       peers stream at once, each over its own xGMI link to the root; a ring all-gather would be bound by one link)
    3. with interleaved sharding, either one scatter per peer into the path-ordered layout (gather_rows: index
       arithmetic on the device, no per-row loop; an int64 index per element, so for node ids and small texts), or no
       copy at all: gather_parts + rows_in_path_order hand a writer the rows in path order where they arrived

Works on HIP tensors over RCCL and on CPU tensors over gloo (the CPU form is what the tests run).

Since round 4 the same exchange also exists BEHIND THE C ABI (include/gbwt_hip.h: gbwt_hip_comm_*, csrc/comm.hip: RCCL called
directly, rows placed in path order by kernels) -- what a Rust `gbunzip` linking libgbwt_hip.so would call; `Comm` below is its
Python face and what bench.py --gpus N and tools/gfa_sharded.py use on GPUs.  The torch.distributed functions stay: they are the
form that runs over gloo on CPU tensors (tests/test_dist_cpu.py) and the fallback when RCCL cannot be loaded by the library.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


def pack_unique_id(uid):
    """All 128 bytes of a gbwt_hip_unique_id (an ncclUniqueId: binary, NULs anywhere), for whatever channel carries it to the other ranks."""
    return C.string_at(C.byref(uid), C.sizeof(uid))


def unpack_unique_id(raw):
    if len(raw) != C.sizeof(_lib.UniqueId):
        raise ValueError(f"unique id of {len(raw)} bytes, expected {C.sizeof(_lib.UniqueId)}")
    uid = _lib.UniqueId()
    C.memmove(C.byref(uid), raw, C.sizeof(uid))
    return uid


class Comm:
    """gbwt_hip_comm: one communicator per process, created collectively by the `world` ranks.  The 128-byte unique id is made by
    rank 0 and reaches the others through `broadcast` -- by default torch.distributed's broadcast_object_list over the default group
    (any backend), or any callable `bytes-or-None -> bytes`."""

    def __init__(self, rank, world, device, broadcast=None):
        self._L = _lib.lib()
        self._c = None
        uid = _lib.UniqueId()
        # A failure of rank 0 (RCCL not loadable -> GBWT_HIP_UNSUPPORTED, the documented fallback case) must be COLLECTIVE: the other
        # ranks are waiting in the broadcast, so rank 0 broadcasts either the id or an error marker, and every rank raises on the marker.
        failure = None
        if rank == 0:
            status = self._L.gbwt_hip_comm_unique_id(C.byref(uid))
            if status != _lib.OK:
                failure = (status, self._L.gbwt_hip_last_error().decode(errors="replace"))
        if world > 1:
            raw = (pack_unique_id(uid) if failure is None else (b"!", failure)) if rank == 0 else None
            if broadcast is None:
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                raw = broadcast(raw)
            if isinstance(raw, tuple):
                failure = raw[1]
            else:
                uid = unpack_unique_id(raw)
        if failure is not None:
            raise _lib.GbwtHipError(*failure)
        self._c = C.c_void_p()
        self.rank, self.world, self.device = rank, world, device
        _lib.check(self._L.gbwt_hip_comm_create(C.byref(uid), rank, world, device, C.byref(self._c)))

    def close(self):
        if getattr(self, "_c", None):
            self._L.gbwt_hip_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather_rows(self, gbwt, root=0, interleaved=False, layout=None):
        """The rows of the last gbwt.extract_device() of every rank, in path order on `root`: a Paths struct (device memory of the
        communicator, valid until its next gather) there, None elsewhere.  layout: GATHER_BLOCKS / GATHER_INTERLEAVED (= interleaved) /
        GATHER_PARTS (every rank holds its stretch of every row: gbwt.extract_part_device(ids, rank, world))."""
        out = _lib.Paths()
        _lib.check(self._L.gbwt_hip_gather_rows(self._c, gbwt._h, gbwt._ws, root, int(interleaved) if layout is None else int(layout), C.byref(out)))
        return out if self.rank == root else None

    def gather_lines(self, gbz, root=0, interleaved=False):
        """The GFA lines of the last gbz.path_lines_device() of every rank, in path order on `root`: a Lines struct there."""
        out = _lib.Lines()
        _lib.check(self._L.gbwt_hip_gather_lines(self._c, gbz._h, gbz._ws, root, int(interleaved), C.byref(out)))
        return out if self.rank == root else None

    def last(self):
        st = _lib.CommStats()
        _lib.check(self._L.gbwt_hip_comm_last(self._c, C.byref(st)))
        return {"ms": st.ms, "bytes": st.bytes, "staged_send": bool(st.staged_send)}


GATHER_BLOCKS, GATHER_INTERLEAVED, GATHER_PARTS = 0, 1, 2     # include/gbwt_hip.h


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank`: sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_ids(ids, rank, world, interleaved=False):
    """The sequence ids this rank walks.  Contiguous blocks keep the gathered output in path order with one copy
    per rank; interleaving (id k -> rank k mod world) balances ragged path lengths."""
    if interleaved:
        return ids[rank::world]
    lo, hi = shard_bounds(len(ids), rank, world)
    return ids[lo:hi]


def _exchange(lengths, values, rows, sizes, dst, group):
    """Step 2: one batch of point-to-point operations.  Returns (len_parts, val_parts) on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = lengths.device
    ops, len_parts, val_parts = [], None, None
    if rank == dst:
        len_parts = [lengths if r == rank else torch.empty(rows[r], dtype=torch.int64, device=device) for r in range(world)]
        val_parts = [values if r == rank else torch.empty(sizes[r], dtype=values.dtype, device=device) for r in range(world)]
        for r in range(world):
            if r == rank:
                continue
            peer = r if group is None else dist.get_global_rank(group, r)
            if rows[r]:
                ops.append(dist.P2POp(dist.irecv, len_parts[r], peer, group))
            if sizes[r]:
                ops.append(dist.P2POp(dist.irecv, val_parts[r], peer, group))
    else:
        peer = dst if group is None else dist.get_global_rank(group, dst)
        if lengths.numel():
            ops.append(dist.P2POp(dist.isend, lengths, peer, group))
        if values.numel():
            ops.append(dist.P2POp(dist.isend, values, peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return len_parts, val_parts


def gather_parts(lengths, values, dst=0, group=None):
    """Steps 1 and 2 only: on `dst` the lists (len_parts, val_parts) with one entry per rank (this rank's own tensors at
    its place, no copy), (None, None) elsewhere.  For payloads that are consumed row by row -- a file writer -- this
    is all that is needed: see rows_in_path_order."""
    world = dist.get_world_size(group)
    device = lengths.device
    counts = torch.tensor([lengths.numel(), values.numel()], dtype=torch.int64, device=device)
    all_counts = torch.zeros(2 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    all_counts = all_counts.cpu().tolist()      # the one host synchronisation: the receive buffers are sized from it
    return _exchange(lengths, values, all_counts[0::2], all_counts[1::2], dst, group)


def join_row_parts(len_parts, val_parts):
    """Every rank holds ITS stretch of every row (api.GBWT.extract_part_device(ids, rank, world)): (offsets int64[n + 1], values) of the
    whole rows -- row k = its stretches in rank order.  The torch form of gbwt_hip_gather_rows(..., GBWT_HIP_GATHER_PARTS, ...)."""
    n = int(len_parts[0].numel())
    assert all(int(p.numel()) == n for p in len_parts), "the ranks do not hold the same rows"
    lens = torch.stack([p.to(torch.int64) for p in len_parts])              # [world, n]
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=lens.device)
    torch.cumsum(lens.sum(0), 0, out=offsets[1:])
    values = torch.empty(int(offsets[-1]), dtype=val_parts[0].dtype, device=val_parts[0].device)
    before = torch.zeros(n, dtype=torch.int64, device=lens.device)          # nodes of row k held by the ranks before r
    for r, part in enumerate(val_parts):
        src_first = torch.cumsum(lens[r], 0) - lens[r]
        shift = torch.repeat_interleave(offsets[:-1] + before - src_first, lens[r])
        values[(torch.arange(int(part.numel()), device=shift.device) + shift).to(values.device)] = part
        before += lens[r]
    return offsets, values


def rows_in_path_order(len_parts):
    """(rank, begin, end) of every gathered row in global path order for interleaved shards (row k of rank r is global
    row k * world + r): the slice val_parts[rank][begin:end] is that row.  Host-side bookkeeping from one copy of the
    lengths; what a writer iterates over instead of building the interleaved array (which would cost an index per
    element -- eight bytes per byte of GFA text)."""
    world = len(len_parts)
    ends = [torch.cumsum(p, 0).cpu().tolist() for p in len_parts]
    rows = max((len(e) for e in ends), default=0)
    for k in range(rows):
        for r in range(world):
            if k < len(ends[r]):
                yield r, (ends[r][k - 1] if k else 0), ends[r][k]


def gather_rows(lengths, values, dst=0, group=None, interleaved=False):
    """Gathers this rank's CSR rows (lengths[k] values each, concatenated in `values`) on rank `dst`.

    Returns (offsets, values) of all rows in global path order on `dst`, (None, None) elsewhere.  `lengths` is an
    int64 tensor, `values` any 1-D tensor; both on the same device.  With `interleaved`, row k of rank r is global
    row k * world + r (the layout of shard_ids(..., interleaved=True))."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = lengths.device
    len_parts, val_parts = gather_parts(lengths, values, dst, group)
    rows = [int(p.numel()) for p in len_parts] if len_parts is not None else None
    sizes = [int(p.numel()) for p in val_parts] if val_parts is not None else None
    if rank != dst:
        return None, None
    if not interleaved:
        all_len = torch.cat(len_parts)
        all_val = torch.cat(val_parts)
        offsets = torch.zeros(all_len.numel() + 1, dtype=torch.int64, device=device)
        torch.cumsum(all_len, 0, out=offsets[1:])
        return offsets, all_val
    total_rows, total = sum(rows), sum(sizes)
    all_len = torch.zeros(total_rows, dtype=torch.int64, device=device)
    for r in range(world):
        all_len[r::world] = len_parts[r]
    offsets = torch.zeros(total_rows + 1, dtype=torch.int64, device=device)
    torch.cumsum(all_len, 0, out=offsets[1:])
    all_val = torch.empty(total, dtype=values.dtype, device=device)
    for r in range(world):
        if sizes[r] == 0:
            continue
        # element e of the peer's row k goes to offsets[k * world + r] + (e - first element of row k)
        src_start = torch.cumsum(len_parts[r], 0) - len_parts[r]
        shift = offsets[r:total_rows:world] - src_start
        where = torch.arange(sizes[r], dtype=torch.int64, device=device) + torch.repeat_interleave(shift, len_parts[r], output_size=sizes[r])
        all_val.index_copy_(0, where, val_parts[r])
    return offsets, all_val


def gather_lines(line_offsets, text, dst=0, group=None, interleaved=False):
    """The final GFA concatenation: this rank's finished lines (`text`, uint8; line k at line_offsets[k] ..
    line_offsets[k + 1], int64) gathered on `dst` in path order.  Returns (offsets, text) there, (None, None) elsewhere.
    With contiguous shards the lines of a rank travel as ONE row (their order is already final).  With interleaved
    shards the dense result costs an index per byte on `dst`: for gigabytes of text use gather_parts + rows_in_path_order
    (tools/gfa_sharded.py)."""
    if interleaved:
        return gather_rows(line_offsets[1:] - line_offsets[:-1], text, dst=dst, group=group, interleaved=True)
    whole = torch.tensor([text.numel()], dtype=torch.int64, device=text.device)
    offsets, all_text = gather_rows(whole, text, dst=dst, group=group)
    return offsets, all_text


class _DeviceMemory:
    """A span of HBM owned by a libgbwt_hip workspace, described the way torch.as_tensor understands."""

    def __init__(self, pointer, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (int(pointer), False), "version": 3}


def device_view(pointer, count, dtype, device):
    """A torch tensor over `count` elements of device memory at `pointer` (no copy).  The memory belongs to the
    workspace that produced it and is valid until the next call on that workspace: clone() what must outlive it."""
    typestr = {torch.uint8: "|u1", torch.int32: "<i4", torch.int64: "<i8"}[dtype]
    if count == 0 or not pointer:
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_DeviceMemory(pointer, count, typestr), device=device)


def lines_tensors(lines, device):
    """(line_offsets int64[n + 1], text uint8[total]) views of a gbwt_hip_lines result (api.GBZ.path_lines_device)."""
    offsets = device_view(lines.d_line_offsets, lines.n + 1 if lines.n else 0, torch.int64, device)   # u64 offsets < 2^63
    text = device_view(lines.d_text, lines.total, torch.uint8, device)
    if lines.n == 0:
        offsets = torch.zeros(1, dtype=torch.int64, device=device)
    return offsets, text


def paths_tensors(paths, device):
    """(offsets int64[n + 1], nodes int32[total]) views of a gbwt_hip_paths result (api.GBWT.extract_device); node ids
    are u32 on the device and < 2^31 whenever alphabet_size is."""
    return device_view(paths.d_offsets, paths.n + 1, torch.int64, device), device_view(paths.d_nodes, paths.total, torch.int32, device)
