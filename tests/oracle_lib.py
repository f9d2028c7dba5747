"""ctypes bindings for the CPU oracle (oracle/liboracle.so).

Test infrastructure only: the oracle is the checker for the HIP path, never the thing shipped.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class Pos(C.Structure):
    _fields_ = [("node", C.c_uint64), ("offset", C.c_uint64)]

    def tup(self):
        return (self.node, self.offset)


class Run(C.Structure):
    _fields_ = [("value", C.c_uint64), ("len", C.c_uint64)]


class State(C.Structure):
    _fields_ = [("node", C.c_uint64), ("start", C.c_uint64), ("end", C.c_uint64)]

    def tup(self):
        return (self.node, self.start, self.end)


class BdState(C.Structure):
    _fields_ = [("forward", State), ("reverse", State)]

    def tup(self):
        return (self.forward.tup(), self.reverse.tup())


class Bytes(C.Structure):
    _fields_ = [("bytes", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]


class RLE(C.Structure):
    _fields_ = [("bytes", Bytes), ("sigma", C.c_uint64), ("threshold", C.c_uint64)]


class RLEIter(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("len", C.c_size_t), ("offset", C.c_size_t),
                ("sigma", C.c_uint64), ("threshold", C.c_uint64)]


class Sparse(C.Structure):
    _fields_ = [("universe", C.c_uint64), ("ones", C.c_uint64), ("high", C.c_void_p), ("high_bits", C.c_uint64),
                ("low", C.c_void_p), ("low_width", C.c_uint64), ("low_len", C.c_uint64), ("samples", C.c_void_p)]


class Record(C.Structure):
    _fields_ = [("id", C.c_uint64), ("edges", C.POINTER(Pos)), ("outdegree", C.c_uint64),
                ("bwt", C.POINTER(C.c_uint8)), ("bwt_len", C.c_size_t)]


class Builder(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("n", C.c_size_t), ("cap", C.c_size_t), ("encoder", RLE)]


_lib = None
LIB_OVERRIDE = None  # bench.py's cpu_baseline leg points this at the -march=native build before first use


def build(target="all"):
    subprocess.check_call(["make", "-C", ORACLE_DIR, target], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_OVERRIDE or os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    u64, p = C.c_uint64, C.c_void_p
    sig = {
        "go_bytes_free": (None, [C.POINTER(Bytes)]),
        "go_bytecode_write": (None, [C.POINTER(Bytes), u64]),
        "go_bytecode_next": (C.c_int, [p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(u64)]),
        "go_rle_init": (None, [C.POINTER(RLE), u64]),
        "go_rle_write": (None, [C.POINTER(RLE), Run]),
        "go_rle_iter_init": (None, [C.POINTER(RLEIter), p, C.c_size_t, u64]),
        "go_rle_iter_next": (C.c_int, [C.POINTER(RLEIter), C.POINTER(Run)]),
        "go_sparse_build": (C.c_int, [C.POINTER(Sparse), u64, p, u64]),
        "go_sparse_free": (None, [C.POINTER(Sparse)]),
        "go_sparse_select": (u64, [C.POINTER(Sparse), u64, C.POINTER(u64)]),
        "go_sparse_next": (u64, [C.POINTER(Sparse), u64, C.POINTER(u64)]),
        "go_builder_init": (None, [C.POINTER(Builder)]),
        "go_builder_append": (None, [C.POINTER(Builder), p, C.c_size_t, p, C.c_size_t]),
        "go_bwt_from_builder": (p, [C.POINTER(Builder)]),
        "go_bwt_from_parts": (p, [p, u64, p, u64]),
        "go_bwt_free": (None, [p]),
        "go_bwt_len": (u64, [p]),
        "go_bwt_data_len": (u64, [p]),
        "go_bwt_data": (p, [p]),
        "go_bwt_record_bytes": (None, [p, u64, C.POINTER(p), C.POINTER(C.c_size_t)]),
        "go_bwt_record": (C.c_int, [p, u64, C.POINTER(Record)]),
        "go_record_free": (None, [C.POINTER(Record)]),
        "go_bwt_compressed_record": (C.c_int64, [p, u64, C.POINTER(p), C.POINTER(C.c_size_t)]),
        "go_record_len": (u64, [C.POINTER(Record)]),
        "go_record_decompress": (C.POINTER(Pos), [C.POINTER(Record), C.POINTER(u64)]),
        "go_record_lf": (C.c_int, [C.POINTER(Record), u64, C.POINTER(Pos)]),
        "go_record_predecessor_at": (C.c_int, [C.POINTER(Record), u64, C.POINTER(u64)]),
        "go_record_offset_to": (C.c_int, [C.POINTER(Record), Pos, C.POINTER(u64)]),
        "go_record_follow": (C.c_int, [C.POINTER(Record), u64, u64, u64, C.POINTER(u64), C.POINTER(u64)]),
        "go_record_bd_follow": (C.c_int, [C.POINTER(Record), u64, u64, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]),
        "go_gbwt_from_bwt": (p, [p, u64, u64, u64, u64, C.c_int]),
        "go_gbwt_free": (None, [p]),
        "go_gbwt_bwt": (p, [p]),
        "go_gbwt_len": (u64, [p]),
        "go_gbwt_sequences": (u64, [p]),
        "go_gbwt_alphabet_size": (u64, [p]),
        "go_gbwt_alphabet_offset": (u64, [p]),
        "go_gbwt_is_bidirectional": (C.c_int, [p]),
        "go_gbwt_start": (C.c_int, [p, u64, C.POINTER(Pos)]),
        "go_gbwt_forward": (C.c_int, [p, Pos, C.POINTER(Pos)]),
        "go_gbwt_backward": (C.c_int, [p, Pos, C.POINTER(Pos)]),
        "go_gbwt_sequence": (C.c_int64, [p, u64, p, u64]),
        "go_gbwt_find": (C.c_int, [p, u64, C.POINTER(State)]),
        "go_gbwt_extend": (C.c_int, [p, C.POINTER(State), u64, C.POINTER(State)]),
        "go_gbwt_bd_find": (C.c_int, [p, u64, C.POINTER(BdState)]),
        "go_gbwt_extend_forward": (C.c_int, [p, C.POINTER(BdState), u64, C.POINTER(BdState)]),
        "go_gbwt_extend_backward": (C.c_int, [p, C.POINTER(BdState), u64, C.POINTER(BdState)]),
        "go_gbwt_extract_mt": (u64, [p, p, u64, C.c_int, p, p, p]),
        "go_gbwt_extract_sums_mt": (u64, [p, p, u64, C.c_int, p, p, p]),
        "go_gbwt_search_mt": (u64, [p, p, u64, u64, C.c_int, p, p]),
        "go_gbwt_bd_search_mt": (u64, [p, p, u64, u64, u64, C.c_int, p, p]),
        "go_gbwt_extract_bytes": (u64, [p, p, u64, C.POINTER(u64)]),
        "go_gbwt_load": (p, [C.c_char_p, C.c_char_p, C.c_size_t]),
        "go_gbz_load": (p, [C.c_char_p, C.c_char_p, C.c_size_t]),
        "go_gbz_free": (None, [p]),
        "go_gbz_gbwt": (p, [p]),
        "go_gbz_write_gfa": (p, [p, C.POINTER(C.c_size_t)]),
        "go_gbz_write_gfa_mode": (p, [p, C.c_int, C.POINTER(C.c_size_t)]),
        "go_metadata_pan_sn_path": (p, [p, u64, C.POINTER(C.c_size_t)]),
        "go_gbz_path_lines": (p, [p, p, u64, C.c_int, C.POINTER(C.c_size_t)]),
        "go_gbz_segment_path": (C.c_int64, [p, u64, p, u64]),
        "go_free": (None, [p]),
        "go_gbz_paths": (u64, [p]),
        "go_gbwt_has_metadata": (C.c_int, [p]),
        "go_gbwt_metadata_paths": (u64, [p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


# ---------------------------------------------------------------------------------------------
# Pythonic helpers


def bytecode_encode(values):
    L = lib()
    b = Bytes()
    for v in values:
        L.go_bytecode_write(C.byref(b), v)
    out = bytes(bytearray(b.bytes[i] for i in range(b.len)))
    L.go_bytes_free(C.byref(b))
    return out


def bytecode_decode(data):
    L = lib()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else (C.c_uint8 * 1)()
    off = C.c_size_t(0)
    val = C.c_uint64(0)
    out = []
    while L.go_bytecode_next(buf, len(data), C.byref(off), C.byref(val)):
        out.append(val.value)
    return out


def rle_encode(sigma, runs):
    L = lib()
    r = RLE()
    L.go_rle_init(C.byref(r), sigma)
    for v, l in runs:
        L.go_rle_write(C.byref(r), Run(v, l))
    out = bytes(bytearray(r.bytes.bytes[i] for i in range(r.bytes.len)))
    L.go_bytes_free(C.byref(r.bytes))
    return out


def rle_decode(sigma, data):
    L = lib()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data.ljust(1, b"\0")) if True else None
    it = RLEIter()
    L.go_rle_iter_init(C.byref(it), buf, len(data), sigma)
    run = Run()
    out = []
    while L.go_rle_iter_next(C.byref(it), C.byref(run)):
        out.append((run.value, run.len))
    return out


class OracleRecord:
    def __init__(self, bwt, i):
        self.L = lib()
        self.rec = Record()
        self.ok = bool(self.L.go_bwt_record(bwt, i, C.byref(self.rec)))

    def __del__(self):
        if self.ok:
            self.L.go_record_free(C.byref(self.rec))
            self.ok = False

    @property
    def outdegree(self):
        return self.rec.outdegree

    def edges(self):
        return [self.rec.edges[k].tup() for k in range(self.rec.outdegree)]

    def bwt_bytes(self):
        return bytes(bytearray(self.rec.bwt[k] for k in range(self.rec.bwt_len)))

    def len(self):
        return self.L.go_record_len(C.byref(self.rec))

    def decompress(self):
        n = C.c_uint64(0)
        ptr = self.L.go_record_decompress(C.byref(self.rec), C.byref(n))
        out = [ptr[k].tup() for k in range(n.value)]
        self.L.go_free(ptr)
        return out

    def lf(self, i):
        out = Pos()
        return out.tup() if self.L.go_record_lf(C.byref(self.rec), i, C.byref(out)) else None

    def predecessor_at(self, i):
        out = C.c_uint64(0)
        return out.value if self.L.go_record_predecessor_at(C.byref(self.rec), i, C.byref(out)) else None

    def offset_to(self, pos):
        out = C.c_uint64(0)
        return out.value if self.L.go_record_offset_to(C.byref(self.rec), Pos(*pos), C.byref(out)) else None

    def follow(self, start, end, node):
        a, b = C.c_uint64(0), C.c_uint64(0)
        if self.L.go_record_follow(C.byref(self.rec), start, end, node, C.byref(a), C.byref(b)):
            return (a.value, b.value)
        return None

    def bd_follow(self, start, end, node):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        if self.L.go_record_bd_follow(C.byref(self.rec), start, end, node, C.byref(a), C.byref(b), C.byref(c)):
            return ((a.value, b.value), c.value)
        return None


class OracleBWT:
    """BWT built like the reference's tests do (BWTBuilder::append, src/bwt/tests.rs:89-101)."""

    def __init__(self, edges=None, runs=None, handle=None, owned=True, parent=None):
        self.L = lib()
        self.owned = owned
        self.parent = parent  # keeps the owning GBWT alive for borrowed views
        if handle is not None:
            self.h = handle
            return
        b = Builder()
        self.L.go_builder_init(C.byref(b))
        for e, r in zip(edges, runs):
            ea = (Pos * max(1, len(e)))(*[Pos(*x) for x in e])
            ra = (Run * max(1, len(r)))(*[Run(*x) for x in r])
            self.L.go_builder_append(C.byref(b), ea, len(e), ra, len(r))
        self.h = self.L.go_bwt_from_builder(C.byref(b))

    @classmethod
    def from_parts(cls, data, starts):
        L = lib()
        d = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        s = np.ascontiguousarray(starts, dtype=np.uint64)
        h = L.go_bwt_from_parts(d.ctypes.data, len(data), s.ctypes.data, len(s))
        assert h, "invalid record starts"
        return cls(handle=h)

    def release(self):
        h, self.h, self.owned = self.h, None, False
        return h

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            self.L.go_bwt_free(self.h)
            self.h = None

    def __len__(self):
        return self.L.go_bwt_len(self.h)

    def data(self):
        n = self.L.go_bwt_data_len(self.h)
        return C.string_at(self.L.go_bwt_data(self.h), n) if n else b""

    def record_bytes(self, i):
        p, n = C.c_void_p(), C.c_size_t(0)
        self.L.go_bwt_record_bytes(self.h, i, C.byref(p), C.byref(n))
        return C.string_at(p, n.value) if n.value else b""

    def starts(self):
        out, off = [], 0
        for i in range(len(self)):
            out.append(off)
            off += len(self.record_bytes(i))
        return out

    def record(self, i):
        r = OracleRecord(self.h, i)
        return r if r.ok else None

    def compressed_record(self, i):
        p, n = C.c_void_p(), C.c_size_t(0)
        off = self.L.go_bwt_compressed_record(self.h, i, C.byref(p), C.byref(n))
        if off < 0:
            return None
        raw = C.string_at(p, n.value)
        return raw[:off], raw[off:]


class OracleGBWT:
    def __init__(self, handle, owner=None):
        self.L = lib()
        self.h = handle
        self.owner = owner  # keeps a GBZ alive

    @classmethod
    def load(cls, path):
        L = lib()
        err = C.create_string_buffer(256)
        h = L.go_gbwt_load(os.fsencode(path), err, 256)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    @classmethod
    def from_bwt(cls, bwt, sequences, size, offset, alphabet_size, bidirectional):
        L = lib()
        h = L.go_gbwt_from_bwt(bwt.release(), sequences, size, offset, alphabet_size, int(bidirectional))
        return cls(h)

    def __del__(self):
        if self.owner is None and self.h:
            self.L.go_gbwt_free(self.h)
            self.h = None

    def bwt(self):
        return OracleBWT(handle=self.L.go_gbwt_bwt(self.h), owned=False, parent=self)

    def len(self):
        return self.L.go_gbwt_len(self.h)

    def sequences(self):
        return self.L.go_gbwt_sequences(self.h)

    def alphabet_size(self):
        return self.L.go_gbwt_alphabet_size(self.h)

    def alphabet_offset(self):
        return self.L.go_gbwt_alphabet_offset(self.h)

    def first_node(self):
        return self.alphabet_offset() + 1

    def is_bidirectional(self):
        return bool(self.L.go_gbwt_is_bidirectional(self.h))

    def has_metadata(self):
        return bool(self.L.go_gbwt_has_metadata(self.h))

    def start(self, i):
        out = Pos()
        return out.tup() if self.L.go_gbwt_start(self.h, i, C.byref(out)) else None

    def forward(self, pos):
        out = Pos()
        return out.tup() if self.L.go_gbwt_forward(self.h, Pos(*pos), C.byref(out)) else None

    def backward(self, pos):
        out = Pos()
        return out.tup() if self.L.go_gbwt_backward(self.h, Pos(*pos), C.byref(out)) == 1 else None

    def sequence(self, i):
        n = self.L.go_gbwt_sequence(self.h, i, None, 0)
        if n < 0:
            return None
        buf = np.zeros(max(1, n), dtype=np.uint64)
        self.L.go_gbwt_sequence(self.h, i, buf.ctypes.data, n)
        return [int(x) for x in buf[:n]]

    def find(self, node):
        out = State()
        return out.tup() if self.L.go_gbwt_find(self.h, node, C.byref(out)) else None

    def extend(self, state, node):
        out = State()
        st = State(*state)
        return out.tup() if self.L.go_gbwt_extend(self.h, C.byref(st), node, C.byref(out)) else None

    @staticmethod
    def _bd(state):
        return BdState(State(*state[0]), State(*state[1]))

    def bd_find(self, node):
        out = BdState()
        return out.tup() if self.L.go_gbwt_bd_find(self.h, node, C.byref(out)) == 1 else None

    def extend_forward(self, state, node):
        out = BdState()
        st = self._bd(state)
        return out.tup() if self.L.go_gbwt_extend_forward(self.h, C.byref(st), node, C.byref(out)) == 1 else None

    def follow(self, state, backward=False):
        """GBZ::follow_forward / follow_backward + StateIter::next (src/gbz.rs:519-544, 1226-1236), stated the way the
        reference's own check_states does (src/gbz/tests.rs:100-168): extend_forward over the successors of the last
        node, skipping the ENDMARKER edge; backward = the same on the flipped state, results flipped back.
        Returns None where GBZ::successors returns None (node without a record)."""
        fwd, rev = state
        if backward:
            fwd, rev = rev, fwd
        node = fwd[0]
        first = self.alphabet_offset() + 1
        if node < first or node >= self.alphabet_size():
            return None
        bwt = self.bwt()
        if bwt.record((node & ~1) - self.alphabet_offset()) is None:     # GBZ::has_node
            return None
        rec = bwt.record(node - self.alphabet_offset())
        if rec is None:
            return None
        out = []
        for succ, _ in rec.edges():
            if succ == 0:
                continue
            ext = self.extend_forward((fwd, rev), succ)
            if ext is not None:
                out.append((ext[1], ext[0]) if backward else ext)
        return out

    def extend_backward(self, state, node):
        out = BdState()
        st = self._bd(state)
        return out.tup() if self.L.go_gbwt_extend_backward(self.h, C.byref(st), node, C.byref(out)) == 1 else None

    def extract(self, seq_ids, threads=1):
        """CSR (offsets, nodes) for the given sequence ids, like gbunzip's worker pool."""
        ids = np.ascontiguousarray(seq_ids, dtype=np.uint64)
        n = len(ids)
        lengths = np.zeros(max(1, n), dtype=np.uint64)
        self.L.go_gbwt_extract_mt(self.h, ids.ctypes.data, n, threads, lengths.ctypes.data, None, None)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(lengths[:n], out=offsets[1:])
        nodes = np.zeros(max(1, int(offsets[-1])), dtype=np.uint32)
        steps = self.L.go_gbwt_extract_mt(self.h, ids.ctypes.data, n, threads, lengths.ctypes.data,
                                          offsets.ctypes.data, nodes.ctypes.data)
        assert steps == int(offsets[-1])
        return offsets, nodes[: int(offsets[-1])]

    def extract_timed(self, seq_ids, threads):
        """Counting-only run (what the cpu_baseline leg times): returns total LF steps."""
        ids = np.ascontiguousarray(seq_ids, dtype=np.uint64)
        return self.L.go_gbwt_extract_mt(self.h, ids.ctypes.data, len(ids), threads, None, None, None)

    def extract_checksums(self, seq_ids, threads):
        """The same walk keeping one (length, sum of node ids, order-dependent hash) per sequence instead of the rows: what bench.py's
        cpu_baseline times AND compares with gbwt_hip_path_sums / gbwt_hip_path_hashes.  Returns (steps, lengths, sums, hashes)."""
        ids = np.ascontiguousarray(seq_ids, dtype=np.uint64)
        n = len(ids)
        lengths, sums, hashes = (np.zeros(max(1, n), dtype=np.uint64) for _ in range(3))
        steps = self.L.go_gbwt_extract_sums_mt(self.h, ids.ctypes.data, n, threads, lengths.ctypes.data, sums.ctypes.data, hashes.ctypes.data)
        return steps, lengths[:n], sums[:n], hashes[:n]

    def search_batch(self, queries, threads=1):
        """find + extend over every row (src/bin/benchmark.rs:155-169); returns (states[n,3] u64, valid[n] bool)."""
        q = np.ascontiguousarray(queries, dtype=np.uint64)
        out = np.zeros((q.shape[0], 3), dtype=np.uint64)
        valid = np.zeros(q.shape[0], dtype=np.uint8)
        self.L.go_gbwt_search_mt(self.h, q.ctypes.data, q.shape[0], q.shape[1], threads, out.ctypes.data, valid.ctypes.data)
        return out, valid.astype(bool)

    def bd_search_batch(self, queries, first, threads=1):
        q = np.ascontiguousarray(queries, dtype=np.uint64)
        out = np.zeros((q.shape[0], 6), dtype=np.uint64)
        valid = np.zeros(q.shape[0], dtype=np.uint8)
        self.L.go_gbwt_bd_search_mt(self.h, q.ctypes.data, q.shape[0], q.shape[1], first, threads, out.ctypes.data, valid.ctypes.data)
        return out, valid.astype(bool)

    def algorithmic_bytes(self, seq_ids):
        ids = np.ascontiguousarray(seq_ids, dtype=np.uint64)
        steps = C.c_uint64(0)
        total = self.L.go_gbwt_extract_bytes(self.h, ids.ctypes.data, len(ids), C.byref(steps))
        return total, steps.value


class OracleGBZ:
    def __init__(self, path):
        self.L = lib()
        err = C.create_string_buffer(256)
        self.h = self.L.go_gbz_load(os.fsencode(path), err, 256)
        if not self.h:
            raise ValueError(err.value.decode())

    def __del__(self):
        if self.h:
            self.L.go_gbz_free(self.h)
            self.h = None

    def gbwt(self):
        return OracleGBWT(self.L.go_gbz_gbwt(self.h), owner=self)

    def paths(self):
        return self.L.go_gbz_paths(self.h)

    def path(self, path_id, reverse=False):
        seq = self.gbwt().sequence(2 * path_id + int(reverse))
        return None if seq is None else [(x // 2, x & 1) for x in seq]

    def gfa(self, path_mode=0):
        """gbunzip's output; path_mode 0 = default, 1 = pan-sn, 2 = ref-only (PathMode, src/bin/gbunzip.rs:63-76)."""
        n = C.c_size_t(0)
        p = self.L.go_gbz_write_gfa_mode(self.h, path_mode, C.byref(n))
        out = C.string_at(p, n.value)
        self.L.go_free(p)
        return out

    def pan_sn_path(self, path_id):
        """Metadata::pan_sn_path (src/gbwt.rs:709-713); None for an id without a name."""
        n = C.c_size_t(0)
        p = self.L.go_metadata_pan_sn_path(self.L.go_gbz_gbwt(self.h), path_id, C.byref(n))
        if not p:
            return None
        out = C.string_at(p, n.value).decode()
        self.L.go_free(p)
        return out

    def segment_path(self, seq_id):
        """GBZ::segment_path(path, orientation) collected for sequence 2 * path + orientation: [(segment id, orientation), ...], or None
        (no translation / no such sequence); src/gbz.rs:477-489, SegmentPathIter 1098-1169."""
        n = self.L.go_gbz_segment_path(self.h, seq_id, None, 0)
        if n < 0:
            return None
        out = np.zeros(max(n, 1), dtype=np.uint64)
        self.L.go_gbz_segment_path(self.h, seq_id, out.ctypes.data, n)
        return [(int(t) >> 1, int(t) & 1) for t in out[:n]]

    def path_lines(self, path_ids, mode):
        ids = np.ascontiguousarray(path_ids, dtype=np.uint64)
        n = C.c_size_t(0)
        p = self.L.go_gbz_path_lines(self.h, ids.ctypes.data, len(ids), mode, C.byref(n))
        out = C.string_at(p, n.value)
        self.L.go_free(p)
        return out
