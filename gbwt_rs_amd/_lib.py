"""Loader + ctypes signatures of libgbwt_hip.so (the C ABI declared in include/gbwt_hip.h).

The shared library is built in-tree by `make -C gbwt_rs_amd/csrc` (see __graft_entry__.build).
There is no Python or CPU fallback: if the library is missing, importing this module's `lib()`
raises, and on a box without a HIP device every compute call returns GBWT_HIP_NO_DEVICE.
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("GBWT_HIP_LIB") or os.path.join(CSRC, "libgbwt_hip.so")   # GBWT_HIP_LIB: A/B runs of two builds (tools/)
HEADER = os.path.join(os.path.dirname(HERE), "include", "gbwt_hip.h")

OPEN_EXTRACT, OPEN_SEARCH, OPEN_GFA, OPEN_ALL = 1, 2, 4, 7     # gbwt_hip_open_*_flags
OK, INVALID_DATA, IO_ERROR, BAD_ARGUMENT, NO_DEVICE, DEVICE_ERROR, CAPACITY, UNSUPPORTED = range(8)
STATUS_NAMES = ["OK", "INVALID_DATA", "IO_ERROR", "BAD_ARGUMENT", "NO_DEVICE", "DEVICE_ERROR", "CAPACITY", "UNSUPPORTED"]


class Pos(C.Structure):
    _fields_ = [("node", C.c_uint64), ("offset", C.c_uint64)]


class State(C.Structure):
    _fields_ = [("node", C.c_uint64), ("start", C.c_uint64), ("end", C.c_uint64)]


class BdState(C.Structure):
    _fields_ = [("forward", State), ("reverse", State)]


class Stats(C.Structure):
    _fields_ = [("size", C.c_uint64), ("sequences", C.c_uint64), ("alphabet_size", C.c_uint64),
                ("alphabet_offset", C.c_uint64), ("records", C.c_uint64), ("data_bytes", C.c_uint64),
                ("paths", C.c_uint64), ("bidirectional", C.c_uint32), ("has_metadata", C.c_uint32),
                ("is_gbz", C.c_uint32), ("has_translation", C.c_uint32), ("max_record_len", C.c_uint64),
                ("max_outdegree", C.c_uint64)]


class OpenTimes(C.Structure):
    _fields_ = [("parse_ms", C.c_double), ("upload_ms", C.c_double), ("sample_ms", C.c_double), ("total_ms", C.c_double),
                ("samples", C.c_uint64), ("checkpoint_walkers", C.c_uint64), ("checkpoint_orphans", C.c_uint64), ("checkpoint_sampling", C.c_uint32),
                ("checkpoint_rounds", C.c_uint32), ("line_sizes_ms", C.c_double)]


class Memory(C.Structure):
    _fields_ = [("index_device_bytes", C.c_uint64), ("index_host_bytes", C.c_uint64), ("workspace_device_bytes", C.c_uint64),
                ("rows_bytes", C.c_uint64), ("text_bytes", C.c_uint64), ("rows_chunks", C.c_uint64)]


class UniqueId(C.Structure):
    _fields_ = [("bytes", C.c_ubyte * 128)]       # binary: NOT c_char (a c_char array reads as a C string and stops at the first NUL)


class CommStats(C.Structure):
    _fields_ = [("ms", C.c_double), ("bytes", C.c_uint64), ("staged_send", C.c_uint32), ("reserved", C.c_uint32)]


class Paths(C.Structure):
    _fields_ = [("d_offsets", C.c_void_p), ("d_nodes", C.c_void_p), ("total", C.c_uint64), ("n", C.c_uint64)]


class States(C.Structure):
    _fields_ = [("d_states", C.c_void_p), ("d_valid", C.c_void_p), ("n", C.c_uint64)]


class Lines(C.Structure):
    _fields_ = [("d_text", C.c_void_p), ("d_line_offsets", C.c_void_p), ("total", C.c_uint64), ("n", C.c_uint64)]


_p, _u64, _int = C.c_void_p, C.c_uint64, C.c_int

SIGNATURES = {
    "gbwt_hip_last_error": (C.c_char_p, []),
    "gbwt_hip_device_count": (_int, []),
    "gbwt_hip_open_file": (_int, [C.c_char_p, _int, C.POINTER(_p)]),
    "gbwt_hip_parse_file": (_int, [C.c_char_p, C.POINTER(Stats)]),
    "gbwt_hip_open_records": (_int, [_p, _u64, _p, _u64, _u64, _u64, _u64, _u64, _int, _int, C.POINTER(_p)]),
    "gbwt_hip_open_file_flags": (_int, [C.c_char_p, _int, C.c_uint32, C.POINTER(_p)]),
    "gbwt_hip_open_records_flags": (_int, [_p, _u64, _p, _u64, _u64, _u64, _u64, _u64, _int, _int, C.c_uint32, C.POINTER(_p)]),
    "gbwt_hip_close": (None, [_p]),
    "gbwt_hip_get_stats": (_int, [_p, C.POINTER(Stats)]),
    "gbwt_hip_get_open_times": (_int, [_p, C.POINTER(OpenTimes)]),
    "gbwt_hip_memory_usage": (_int, [_p, _p, C.POINTER(Memory)]),
    "gbwt_hip_last_lines_ms": (_int, [_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gbwt_hip_workspace_create": (_int, [_p, C.POINTER(_p)]),
    "gbwt_hip_workspace_destroy": (None, [_p]),
    "gbwt_hip_workspace_stream": (_p, [_p]),
    "gbwt_hip_workspace_tune": (_int, [_p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "gbwt_hip_extract": (_int, [_p, _p, _p, _u64, _p, _p, _u64, C.POINTER(_u64)]),
    "gbwt_hip_extract_device": (_int, [_p, _p, _p, _u64, C.POINTER(Paths)]),
    "gbwt_hip_extract_part_device": (_int, [_p, _p, _p, _u64, C.c_uint32, C.c_uint32, C.POINTER(Paths)]),
    "gbwt_hip_copy_result": (_int, [_p, _p, _p, _p, _u64]),
    "gbwt_hip_extract_paths": (_int, [_p, _p, _p, _u64, _int, _p, _p, _u64, C.POINTER(_u64)]),
    "gbwt_hip_start": (_int, [_p, _p, _p, _u64, _p, _p]),
    "gbwt_hip_forward": (_int, [_p, _p, _p, _u64, _p, _p]),
    "gbwt_hip_backward": (_int, [_p, _p, _p, _u64, _p, _p]),
    "gbwt_hip_find": (_int, [_p, _p, _p, _u64, _p, _p]),
    "gbwt_hip_extend": (_int, [_p, _p, _p, _p, _u64, _p, _p]),
    "gbwt_hip_bd_find": (_int, [_p, _p, _p, _u64, _p, _p]),
    "gbwt_hip_extend_forward": (_int, [_p, _p, _p, _p, _u64, _p, _p]),
    "gbwt_hip_follow": (_int, [_p, _p, _p, _u64, _int, _p, _p, _u64, _p, _p]),
    "gbwt_hip_extend_backward": (_int, [_p, _p, _p, _p, _u64, _p, _p]),
    "gbwt_hip_search": (_int, [_p, _p, _p, _u64, _u64, _p, _p]),
    "gbwt_hip_bd_search": (_int, [_p, _p, _p, _u64, _u64, _u64, _p, _p]),
    "gbwt_hip_search_device": (_int, [_p, _p, _p, _u64, _u64, C.POINTER(States)]),
    "gbwt_hip_bd_search_device": (_int, [_p, _p, _p, _u64, _u64, _u64, C.POINTER(States)]),
    "gbwt_hip_path_lines": (_int, [_p, _p, _p, _u64, _int, _p, _u64, C.POINTER(_u64)]),
    "gbwt_hip_path_lines_device": (_int, [_p, _p, _p, _u64, _int, C.POINTER(Lines)]),
    "gbwt_hip_segment_paths": (_int, [_p, _p, _p, _u64, _p, _p, _u64, C.POINTER(_u64)]),
    "gbwt_hip_write_gfa": (_int, [_p, _p, C.c_char_p]),
    "gbwt_hip_write_gfa_mode": (_int, [_p, _p, C.c_char_p, _int]),
    "gbwt_hip_path_sums": (_int, [_p, _p, _p, _u64]),
    "gbwt_hip_path_hashes": (_int, [_p, _p, _p, _u64]),
    "gbwt_hip_copy_path": (_int, [_p, _p, _u64, _p, _u64, C.POINTER(_u64)]),
    "gbwt_hip_last_kernel_ms": (_int, [_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gbwt_hip_last_query_ms": (_int, [_p, C.POINTER(C.c_float)]),
    "gbwt_hip_device_memory": (_int, [_int, C.POINTER(_u64), C.POINTER(_u64)]),
    "gbwt_hip_comm_unique_id": (_int, [C.POINTER(UniqueId)]),
    "gbwt_hip_comm_create": (_int, [C.POINTER(UniqueId), _int, _int, _int, C.POINTER(_p)]),
    "gbwt_hip_comm_destroy": (None, [_p]),
    "gbwt_hip_gather_rows": (_int, [_p, _p, _p, _int, _int, C.POINTER(Paths)]),
    "gbwt_hip_gather_lines": (_int, [_p, _p, _p, _int, _int, C.POINTER(Lines)]),
    "gbwt_hip_comm_last": (_int, [_p, C.POINTER(CommStats)]),
}

_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C {CSRC}` (or __graft_entry__.build()); "
                              "gbwt_rs_amd has no CPU fallback")
        # A process that uses torch AND this library must load torch FIRST: torch brings its own copy of the HIP runtime, and when
        # libgbwt_hip.so (linked against /opt/rocm's) has started the device before torch is imported, torch's runtime finds "No HIP
        # GPUs" (measured on the GPU box, round 5: a late `import torch` + .cuda() after the first extraction).  The other order works:
        # this library then binds to the runtime that is already in the process.  So the Python mirror imports torch here, before the
        # library is loaded, when torch is installed (dist.py needs it anyway); GBWT_HIP_NO_TORCH_PRELOAD=1 skips that.
        if "torch" not in sys.modules and not os.environ.get("GBWT_HIP_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:  # noqa: BLE001  (not installed, or installed and broken -- OSError / RuntimeError from its own loader: this
                pass           # library does not need it; CPU-only callers such as gbwt_hip_parse_file must not fail because of it)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class GbwtHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES[status] if 0 <= status < len(STATUS_NAMES) else status}: {message}")
        self.status = status


def check(status):
    if status != OK:
        raise GbwtHipError(status, lib().gbwt_hip_last_error().decode(errors="replace"))
