"""Host-side mirror of the reference's GBWT / GBZ interface for the hot path, batched, on MI355X.

Method names, argument meaning and None-behaviour follow the Rust API (file:line into the reference):

  GBWT.len / sequences / alphabet_size / alphabet_offset / first_node / is_bidirectional   src/gbwt.rs:108-174
  GBWT.start(ids)            GBWT::start            src/gbwt.rs:213-219
  GBWT.forward(positions)    GBWT::forward          src/gbwt.rs:222-229
  GBWT.backward(positions)   GBWT::backward         src/gbwt.rs:236-250
  GBWT.sequence(id)          GBWT::sequence         src/gbwt.rs:253-261   (None for id >= sequences)
  GBWT.sequences_csr(ids)    the batched form of sequence(): CSR arrays
  GBWT.find / extend / bd_find / extend_forward / extend_backward     src/gbwt.rs:269-367
  GBZ.path(path_id, orientation)   GBZ::path        src/gbz.rs:461-466

Every call goes through the C ABI of libgbwt_hip.so (hand-written HIP); "not found" is reported as
None / a False entry of the validity mask, never as an exception.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import BdState, GbwtHipError, Lines, Memory, OpenTimes, Paths, Pos, State, Stats, check

FORWARD, REVERSE = 0, 1  # support::Orientation, src/support.rs:30-47
PATHS_DEFAULT, PATHS_PAN_SN, PATHS_REF_ONLY = 0, 1, 2  # gbunzip's PathMode, src/bin/gbunzip.rs:63-76

POS_DTYPE = np.dtype([("node", "<u8"), ("offset", "<u8")])
STATE_DTYPE = np.dtype([("node", "<u8"), ("start", "<u8"), ("end", "<u8")])
BD_DTYPE = np.dtype([("forward", STATE_DTYPE), ("reverse", STATE_DTYPE)])


def encode_node(node_id, orientation):  # support::encode_node, src/support.rs:155-157
    return 2 * node_id + orientation


def decode_node(node):  # support::decode_node, src/support.rs:180-182
    return node // 2, node & 1


def flip_node(node):  # support::flip_node, src/support.rs:188-190
    return node ^ 1


def encode_path(path_id, orientation):  # support::encode_path, src/support.rs:229-231
    return 2 * path_id + orientation


def device_count():
    return _lib.lib().gbwt_hip_device_count()


def device_memory(device=0):
    """(free, total) bytes of a device (hipMemGetInfo)."""
    free, total = C.c_uint64(0), C.c_uint64(0)
    check(_lib.lib().gbwt_hip_device_memory(device, C.byref(free), C.byref(total)))
    return free.value, total.value


def parse_file(path):
    """Host-only parse + validation (no GPU): the statistics serialize::load_from would yield."""
    st = Stats()
    check(_lib.lib().gbwt_hip_parse_file(os.fsencode(path), C.byref(st)))
    return st


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


class GBWT:
    """A GBWT index resident in HBM.  Mirrors gbwt::GBWT (src/gbwt.rs:95-385) for the hot path."""

    def __init__(self, handle, owner=None, device=0):
        self._L = _lib.lib()
        self._h = handle
        self._device = device
        self._owner = owner                 # a view of another object's index (another_workspace): that object closes it
        self._ws = C.c_void_p()
        check(self._L.gbwt_hip_workspace_create(self._h, C.byref(self._ws)))
        self._stats = Stats()
        check(self._L.gbwt_hip_get_stats(self._h, C.byref(self._stats)))

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def load(cls, path, device=0, flags=_lib.OPEN_ALL):
        """serialize::load_from::<GBWT | GBZ>(path) (src/gbwt.rs:402-438, src/gbz.rs:674-717).  flags: what the handle is for
        (_lib.OPEN_EXTRACT | OPEN_SEARCH | OPEN_GFA, gbwt_hip_open_file_flags): only those structures are built in HBM."""
        h = C.c_void_p()
        check(_lib.lib().gbwt_hip_open_file_flags(os.fsencode(path), device, flags, C.byref(h)))
        return cls(h, device=device)

    @classmethod
    def from_records(cls, data, starts, alphabet_offset, alphabet_size, sequences, size, bidirectional=True, device=0, flags=_lib.OPEN_ALL):
        """From the raw record stream of a bwt::BWT (compressed_record, src/bwt.rs:134-143) + header fields."""
        d = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        s = np.ascontiguousarray(starts, dtype=np.uint64)
        h = C.c_void_p()
        check(_lib.lib().gbwt_hip_open_records_flags(_ptr(d), d.size, _ptr(s), s.size, alphabet_offset, alphabet_size,
                                                     sequences, size, int(bidirectional), device, flags, C.byref(h)))
        return cls(h, device=device)

    def close(self):
        if getattr(self, "_ws", None):
            self._L.gbwt_hip_workspace_destroy(self._ws)
            self._ws = None
        if getattr(self, "_h", None):
            if getattr(self, "_owner", None) is None:
                self._L.gbwt_hip_close(self._h)
            self._h = None
            self._owner = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def another_workspace(self):
        """The same index through a workspace of its own: what a second host thread uses (a handle is immutable after open and safe
        for concurrent read-only calls as long as every thread has its own workspace; the reference shares &GBZ across rayon
        workers, src/bin/gbunzip.rs:421-434).  The view keeps this object alive and never closes the index."""
        return type(self)(self._h, owner=self, device=self._device)

    def new_workspace(self):
        """A fresh workspace: the GBWT_HIP_* extraction knobs are read when a workspace is created, not per call."""
        if getattr(self, "_ws", None):
            self._L.gbwt_hip_workspace_destroy(self._ws)
        self._ws = C.c_void_p()
        check(self._L.gbwt_hip_workspace_create(self._h, C.byref(self._ws)))

    def open_times(self):
        """Where the time of the open went (gbwt_hip_get_open_times): a dict of milliseconds and counts."""
        t = OpenTimes()
        check(self._L.gbwt_hip_get_open_times(self._h, C.byref(t)))
        return {name: getattr(t, name) for name, _ in OpenTimes._fields_}

    def memory_usage(self):
        """Bytes held by the index (device, host) and by this object's workspace (gbwt_hip_memory_usage): a dict."""
        m = Memory()
        check(self._L.gbwt_hip_memory_usage(self._h, self._ws, C.byref(m)))
        return {name: getattr(m, name) for name, _ in Memory._fields_}

    # ---- statistics (src/gbwt.rs:105-175) -----------------------------------------------------
    def len(self):
        return self._stats.size

    def is_empty(self):
        return self.len() == 0

    def sequences(self):
        return self._stats.sequences

    def alphabet_size(self):
        return self._stats.alphabet_size

    def alphabet_offset(self):
        return self._stats.alphabet_offset

    def effective_size(self):
        return self.alphabet_size() - self.alphabet_offset()

    def first_node(self):
        return self.alphabet_offset() + 1

    def has_node(self, node):
        return self.alphabet_offset() < node < self.alphabet_size()

    def is_bidirectional(self):
        return bool(self._stats.bidirectional)

    def has_metadata(self):
        return bool(self._stats.has_metadata)

    @property
    def stats(self):
        return self._stats

    # ---- navigation ---------------------------------------------------------------------------
    def start(self, ids):
        """GBWT::start for an array of sequence ids -> (positions[POS_DTYPE], valid[bool])."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        out = np.zeros(ids.size, dtype=POS_DTYPE)
        valid = np.zeros(ids.size, dtype=np.uint8)
        check(self._L.gbwt_hip_start(self._h, self._ws, _ptr(ids), ids.size, _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    def forward(self, positions):
        """GBWT::forward for an array of positions -> (positions, valid)."""
        pos = np.ascontiguousarray(positions, dtype=POS_DTYPE)
        out = np.zeros(pos.size, dtype=POS_DTYPE)
        valid = np.zeros(pos.size, dtype=np.uint8)
        check(self._L.gbwt_hip_forward(self._h, self._ws, _ptr(pos), pos.size, _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    def backward(self, positions):
        """GBWT::backward (src/gbwt.rs:236-250) for an array of positions -> (positions, valid)."""
        pos = np.ascontiguousarray(positions, dtype=POS_DTYPE)
        out = np.zeros(pos.size, dtype=POS_DTYPE)
        valid = np.zeros(pos.size, dtype=np.uint8)
        check(self._L.gbwt_hip_backward(self._h, self._ws, _ptr(pos), pos.size, _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    def sequences_csr(self, ids, return_valid=False):
        """GBWT::sequence(id).collect() for every id -> (offsets[u64, n+1], nodes[u32]) [, valid[bool]].

        One walk of the sequences: they are extracted into HBM, the host array is sized from the result and filled by a
        copy.  An id >= sequences() -- GBWT::sequence returns None, src/gbwt.rs:254-256 -- gets an empty row and
        valid[k] = False; it never fails the batch."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        out = self.extract_device(ids)
        offsets = np.zeros(ids.size + 1, dtype=np.uint64)
        nodes = np.empty(max(1, out.total), dtype=np.uint32)
        check(self._L.gbwt_hip_copy_result(self._h, self._ws, _ptr(offsets), _ptr(nodes) if out.total else None, nodes.size))
        if return_valid:
            return offsets, nodes[: out.total], ids < np.uint64(self.sequences())
        return offsets, nodes[: out.total]

    def sequence(self, seq_id):
        """GBWT::sequence(id): list of GBWT nodes, or None if there is no such sequence."""
        if seq_id >= self.sequences():
            return None
        offsets, nodes = self.sequences_csr([seq_id])
        return [int(x) for x in nodes]

    def extract_device(self, ids):
        """Device-resident extraction (what bench.py times): returns a Paths struct with device pointers."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        out = Paths()
        check(self._L.gbwt_hip_extract_device(self._h, self._ws, _ptr(ids), ids.size, C.byref(out)))
        return out

    def extract_part_device(self, ids, part, parts):
        """Stretch `part` of `parts` of every row (gbwt_hip_extract_part_device): the rows are cut at sequence samples, the stretches of a
        row back to back are the row.  What one rank of `parts` extracts (dist.Comm.gather_rows(..., layout=GATHER_PARTS) joins them)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        out = Paths()
        check(self._L.gbwt_hip_extract_part_device(self._h, self._ws, _ptr(ids), ids.size, part, parts, C.byref(out)))
        return out

    def part_csr(self, ids, part, parts):
        """extract_part_device + copy: (offsets[u64, n + 1], nodes[u32]) of the stretches."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        out = self.extract_part_device(ids, part, parts)
        offsets = np.zeros(ids.size + 1, dtype=np.uint64)
        nodes = np.empty(max(1, out.total), dtype=np.uint32)
        check(self._L.gbwt_hip_copy_result(self._h, self._ws, _ptr(offsets), _ptr(nodes) if out.total else None, nodes.size))
        return offsets, nodes[: out.total]

    def last_offsets(self, n):
        """The row offsets (u64[n + 1]) of the last extract_device() / extract_part_device() of n rows."""
        offsets = np.zeros(n + 1, dtype=np.uint64)
        check(self._L.gbwt_hip_copy_result(self._h, self._ws, _ptr(offsets), None, 0))
        return offsets

    def path_sums(self, n):
        """Per-path sums of node ids of the last extract_device() (device-side reduction)."""
        out = np.zeros(n, dtype=np.uint64)
        check(self._L.gbwt_hip_path_sums(self._h, self._ws, _ptr(out), n))
        return out

    def path_hashes(self, n):
        """Per-path ORDER-dependent checksums of the last extract_device(): sum of (node + 1) * splitmix64(position), gbwt_hip.h."""
        out = np.zeros(n, dtype=np.uint64)
        check(self._L.gbwt_hip_path_hashes(self._h, self._ws, _ptr(out), n))
        return out

    def copy_path(self, k):
        """Row k of the last extract_device() as a host array."""
        ln = C.c_uint64(0)
        check(self._L.gbwt_hip_copy_path(self._h, self._ws, k, None, 0, C.byref(ln)))
        out = np.zeros(max(1, ln.value), dtype=np.uint32)
        check(self._L.gbwt_hip_copy_path(self._h, self._ws, k, _ptr(out), out.size, C.byref(ln)))
        return out[: ln.value]

    def last_kernel_ms(self):
        walk, total = C.c_float(0), C.c_float(0)
        check(self._L.gbwt_hip_last_kernel_ms(self._ws, C.byref(walk), C.byref(total)))
        return walk.value, total.value

    def last_query_ms(self):
        """Kernel time (HIP events) of the last start / forward / backward / find / extend / search / follow call."""
        ms = C.c_float(0)
        check(self._L.gbwt_hip_last_query_ms(self._ws, C.byref(ms)))
        return ms.value

    def tune(self, walk_mode=0, paths_per_wave=0, small_record=16):
        """Kernel tuning knobs of gbwt_hip_workspace_tune (results never depend on them)."""
        check(self._L.gbwt_hip_workspace_tune(self._ws, walk_mode, paths_per_wave, small_record))

    def stream(self):
        return self._L.gbwt_hip_workspace_stream(self._ws)

    # ---- search -------------------------------------------------------------------------------
    def find(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        out = np.zeros(nodes.size, dtype=STATE_DTYPE)
        valid = np.zeros(nodes.size, dtype=np.uint8)
        check(self._L.gbwt_hip_find(self._h, self._ws, _ptr(nodes), nodes.size, _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    def extend(self, states, nodes):
        states = np.ascontiguousarray(states, dtype=STATE_DTYPE)
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        assert states.size == nodes.size
        out = np.zeros(nodes.size, dtype=STATE_DTYPE)
        valid = np.zeros(nodes.size, dtype=np.uint8)
        check(self._L.gbwt_hip_extend(self._h, self._ws, _ptr(states), _ptr(nodes), nodes.size, _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    def bd_find(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        out = np.zeros(nodes.size, dtype=BD_DTYPE)
        valid = np.zeros(nodes.size, dtype=np.uint8)
        check(self._L.gbwt_hip_bd_find(self._h, self._ws, _ptr(nodes), nodes.size, _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    def _bd_extend(self, fn, states, nodes):
        states = np.ascontiguousarray(states, dtype=BD_DTYPE)
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        assert states.size == nodes.size
        out = np.zeros(nodes.size, dtype=BD_DTYPE)
        valid = np.zeros(nodes.size, dtype=np.uint8)
        check(fn(self._h, self._ws, _ptr(states), _ptr(nodes), nodes.size, _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    def extend_forward(self, states, nodes):
        return self._bd_extend(self._L.gbwt_hip_extend_forward, states, nodes)

    def extend_backward(self, states, nodes):
        return self._bd_extend(self._L.gbwt_hip_extend_backward, states, nodes)

    def follow(self, states, backward=False):
        """GBZ::follow_forward / follow_backward (src/gbz.rs:519-544) for an array of bidirectional states:
        (offsets[n + 1], extensions, valid); valid[i] is False where the reference returns no iterator."""
        st = np.ascontiguousarray(states, dtype=BD_DTYPE)
        offsets = np.zeros(st.size + 1, dtype=np.uint64)
        valid = np.zeros(st.size, dtype=np.uint8)
        total = C.c_uint64(0)
        check(self._L.gbwt_hip_follow(self._h, self._ws, _ptr(st), st.size, int(backward), _ptr(offsets), None, 0, C.byref(total), _ptr(valid)))
        out = np.zeros(max(total.value, 1), dtype=BD_DTYPE)
        check(self._L.gbwt_hip_follow(self._h, self._ws, _ptr(st), st.size, int(backward), _ptr(offsets), _ptr(out), out.size, C.byref(total), _ptr(valid)))
        return offsets, out[:total.value], valid.astype(bool)

    @staticmethod
    def _results(n, dtype, out):
        """Result arrays of a host-pointer query call: fresh ones, or the caller's `out` = (states, valid uint8) of the right shape -- a caller
        that asks again and again keeps its arrays: the device-to-host copy into pages that have been touched takes half the time of the copy
        into fresh ones (profiles/r06_download_probe.txt)."""
        if out is None:
            return np.zeros(n, dtype=dtype), np.zeros(n, dtype=np.uint8)
        states, valid = out
        assert states.dtype == dtype and states.shape == (n,) and states.flags.c_contiguous and valid.dtype == np.uint8 and valid.shape == (n,) and valid.flags.c_contiguous
        return states, valid

    def search(self, queries, out=None):
        """find(q[0]) + extend over q[1:] for every row of the (n, len) query matrix (src/bin/benchmark.rs:155-169).  `out`: see _results."""
        q = np.ascontiguousarray(queries, dtype=np.uint64)
        assert q.ndim == 2
        out, valid = self._results(q.shape[0], STATE_DTYPE, out)
        check(self._L.gbwt_hip_search(self._h, self._ws, _ptr(q), q.shape[0], q.shape[1], _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    def search_device(self, d_queries, n, length):
        """gbwt_hip_search_device: `d_queries` = device pointer (int) to an n x length u64 matrix in HBM; the final states stay in the
        workspace: a _lib.States struct (d_states, d_valid, n).  states_to_host() copies them out."""
        out = _lib.States()
        check(self._L.gbwt_hip_search_device(self._h, self._ws, C.c_void_p(int(d_queries)), n, length, C.byref(out)))
        return out

    def states_to_host(self, states, bidirectional=False):
        """(states, valid) of a search_device() / bd_search_device() result as host arrays (through torch views of the workspace's memory)."""
        import torch
        from . import dist as D
        dtype = BD_DTYPE if bidirectional else STATE_DTYPE
        device = torch.device("cuda", self._device)
        raw = D.device_view(states.d_states, states.n * dtype.itemsize, torch.uint8, device).cpu().numpy()
        valid = D.device_view(states.d_valid, states.n, torch.uint8, device).cpu().numpy()
        return raw.view(dtype).copy(), valid.astype(bool)

    def bd_search_device(self, d_queries, n, length, first):
        out = _lib.States()
        check(self._L.gbwt_hip_bd_search_device(self._h, self._ws, C.c_void_p(int(d_queries)), n, length, first, C.byref(out)))
        return out

    def bd_search(self, queries, first, out=None):
        """bd_find(q[first]) then alternating extend_forward / extend_backward over every row of the query matrix.  `out`: see _results."""
        q = np.ascontiguousarray(queries, dtype=np.uint64)
        assert q.ndim == 2
        out, valid = self._results(q.shape[0], BD_DTYPE, out)
        check(self._L.gbwt_hip_bd_search(self._h, self._ws, _ptr(q), q.shape[0], q.shape[1], first, _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)


class GBZ(GBWT):
    """gbz::GBZ for the hot path: GBZ::path / paths (src/gbz.rs:446-466)."""

    def paths(self):
        return self.sequences() // 2

    def path(self, path_id, orientation=FORWARD):
        """GBZ::path(path_id, orientation): list of (node_id, orientation), or None (src/gbz.rs:461-466)."""
        seq = self.sequence(encode_path(path_id, orientation))
        return None if seq is None else [decode_node(x) for x in seq]

    def path_lines(self, path_ids, mode):
        """gbunzip's P-lines (mode 0), W-lines (mode 1) or P-lines with PanSN names (mode 2) for the given paths, as bytes
        (src/bin/gbunzip.rs:438-550)."""
        ids = np.ascontiguousarray(path_ids, dtype=np.uint64)
        total = C.c_uint64(0)
        check(self._L.gbwt_hip_path_lines(self._h, self._ws, _ptr(ids), ids.size, mode, None, 0, C.byref(total)))
        buf = C.create_string_buffer(max(1, total.value))
        check(self._L.gbwt_hip_path_lines(self._h, self._ws, _ptr(ids), ids.size, mode, buf, total.value, C.byref(total)))
        return buf.raw[: total.value]

    def path_lines_array(self, path_ids, mode, out=None):
        """The same lines as a numpy uint8 array -- into `out` when it is large enough (a reused buffer has its pages already) --
        without the zero-filled ctypes buffer and the second copy `path_lines` pays for a bytes object."""
        ids = np.ascontiguousarray(path_ids, dtype=np.uint64)
        total = C.c_uint64(0)
        check(self._L.gbwt_hip_path_lines(self._h, self._ws, _ptr(ids), ids.size, mode, None, 0, C.byref(total)))
        if out is None or out.size < total.value:
            out = np.empty(max(1, total.value), dtype=np.uint8)
        check(self._L.gbwt_hip_path_lines(self._h, self._ws, _ptr(ids), ids.size, mode, out.ctypes.data, out.size, C.byref(total)))
        return out[: total.value]

    def last_lines_ms(self):
        """(walk kernel ms, everything behind the walk ms) of the last path_lines / path_lines_device request (HIP events)."""
        walk, fmt = C.c_float(0), C.c_float(0)
        check(self._L.gbwt_hip_last_lines_ms(self._ws, C.byref(walk), C.byref(fmt)))
        return walk.value, fmt.value

    def write_gfa(self, path, path_mode=PATHS_DEFAULT):
        """The file `gbunzip -t 1 --paths MODE` writes for this GBZ (src/bin/gbunzip.rs:205-226; PathMode 63-76)."""
        check(self._L.gbwt_hip_write_gfa_mode(self._h, self._ws, os.fsencode(path), path_mode))

    def paths_csr(self, path_ids, orientation=FORWARD):
        """GBZ::path(id, orientation) for every id, as CSR of GBWT-encoded nodes (support::encode_path, src/support.rs:229-231)."""
        ids = np.ascontiguousarray(path_ids, dtype=np.uint64)
        return self.sequences_csr(2 * ids + np.uint64(1 if orientation else 0))

    def segment_paths(self, seq_ids):
        """GBZ::segment_path (src/gbz.rs:477-489) for a batch of SEQUENCE ids (2 * path + orientation): (offsets[n + 1], tokens) with token =
        (segment id << 1) | orientation -- the (Segment, Orientation) pairs SegmentPathIter yields, up to the place where it stops for a path
        that is not a concatenation of whole segments.  GbwtHipError(BAD_ARGUMENT) without a node-to-segment translation (the reference: None)."""
        ids = np.ascontiguousarray(seq_ids, dtype=np.uint64)
        offsets = np.zeros(ids.size + 1, dtype=np.uint64)
        total = C.c_uint64(0)
        check(self._L.gbwt_hip_segment_paths(self._h, self._ws, _ptr(ids), ids.size, _ptr(offsets), None, 0, C.byref(total)))
        tokens = np.zeros(max(total.value, 1), dtype=np.uint64)
        check(self._L.gbwt_hip_segment_paths(self._h, self._ws, _ptr(ids), ids.size, _ptr(offsets), _ptr(tokens), tokens.size, C.byref(total)))
        return offsets, tokens[:total.value]

    def segment_path(self, path_id, orientation=FORWARD):
        """[(segment id, orientation), ...] of one path (GBZ::segment_path(path_id, orientation))."""
        _, tokens = self.segment_paths([2 * path_id + (1 if orientation else 0)])
        return [(int(t) >> 1, int(t) & 1) for t in tokens]

    def path_lines_device(self, path_ids, mode):
        """The same lines left in HBM: a Lines struct (device pointers to the text and to the n + 1 line offsets)."""
        ids = np.ascontiguousarray(path_ids, dtype=np.uint64)
        out = Lines()
        check(self._L.gbwt_hip_path_lines_device(self._h, self._ws, _ptr(ids), ids.size, mode, C.byref(out)))
        return out


__all__ = ["GBWT", "GBZ", "GbwtHipError", "Lines", "Paths", "FORWARD", "REVERSE", "PATHS_DEFAULT", "PATHS_PAN_SN", "PATHS_REF_ONLY", "POS_DTYPE", "STATE_DTYPE", "BD_DTYPE", "encode_node",
           "decode_node", "flip_node", "encode_path", "device_count", "device_memory", "parse_file", "Pos", "State", "BdState"]
