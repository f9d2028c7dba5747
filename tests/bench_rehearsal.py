#!/usr/bin/env python3
"""TEST HARNESS (tests/test_dist_cpu.py: test_bench_gpus2_over_gloo): bench.py's N > 1 flow, end to end, on a machine WITHOUT a GPU.

The first real `bench.py --gpus 8` run must not fail on plumbing -- a collective that one rank skips, a key the line lacks, a rank that
raises while its peers wait.  None of that needs a GPU to find: this script is started once per rank by torch.distributed.run
(gloo, CPU tensors), puts a stand-in for the `gbwt_rs_amd` handle classes in place -- every extraction and every GFA line answered by the
CPU ORACLE (tests/oracle_lib.py), rows "cut" for the parts of gbwt_hip_extract_part_device -- and then runs bench.main() unchanged.
Nothing here is a product path and nothing of it is measured: the line it prints says "rehearsal".  What the stand-in does NOT cover is
what needs hardware: the HIP kernels (pytest -m gpu) and RCCL itself.

REHEARSAL_FAIL=c4_open:<rank> makes that rank's open of config 4's index raise (the ranks must agree on the failure and the headline's
line must still come out); REHEARSAL_FAIL=c4_generate makes rank 0's generator raise; REHEARSAL_FAIL=gather_hang:<rank> makes that rank never
come back from the exchange of the final gather (what a transport that hangs between GPUs would look like: every rank's guard must abandon
the step -- BENCH_GUARD_SECONDS -- and rank 0 must still print the line)."""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import oracle_lib as O  # noqa: E402
import gbwt_rs_amd as G  # noqa: E402  (the package itself: constants, synth, dist -- only the handle classes are replaced)
from gbwt_rs_amd import dist as D  # noqa: E402

FAIL = os.environ.get("REHEARSAL_FAIL", "")
RANK = int(os.environ.get("RANK", "0"))


class _Rows:
    def __init__(self, offsets, nodes):
        self.offsets, self.nodes = offsets.astype(np.uint64), nodes.astype(np.uint32)
        self.total, self.n = int(offsets[-1]), len(offsets) - 1


class _Lines:
    def __init__(self, lines):
        self.text = np.frombuffer(b"".join(lines), dtype=np.uint8).copy() if lines else np.zeros(0, dtype=np.uint8)
        self.offsets = np.concatenate([[0], np.cumsum([len(x) for x in lines])]).astype(np.int64)
        self.total, self.n = int(self.offsets[-1]), len(lines)


class _Stats:
    def __init__(self, gbwt):
        self.data_bytes, self.records = 0, int(gbwt.alphabet_size() - gbwt.alphabet_offset()) if hasattr(gbwt, "alphabet_size") else 0


class FakeIndex:
    """What bench.py and tools/c4_bench.py call on a handle, answered by the oracle."""
    def __init__(self, path=None, gbz=None):
        self._gbz = gbz if gbz is not None else O.OracleGBZ(path)
        self._gbwt = self._gbz.gbwt()
        self._last = None
        self._device = 0

    # ---- handle
    def close(self):
        pass

    def another_workspace(self):
        return FakeIndex(gbz=self._gbz)

    def open_times(self):
        return {"parse_ms": 1.0, "upload_ms": 1.0, "sample_ms": 1.0, "total_ms": 3.0, "samples": 1, "checkpoint_walkers": 1, "checkpoint_orphans": 0,
                "checkpoint_sampling": 1, "checkpoint_rounds": 1, "line_sizes_ms": 0.0}

    def memory_usage(self):
        return {"index_device_bytes": 0, "index_host_bytes": 0, "workspace_device_bytes": 0, "rows_bytes": 0, "text_bytes": 0, "rows_chunks": 0}

    @property
    def stats(self):
        return _Stats(self._gbwt)

    def paths(self):
        return self._gbwt.sequences() // 2

    def len(self):
        return self._gbwt.len()

    def sequences(self):
        return self._gbwt.sequences()

    # ---- extraction
    def extract_device(self, ids):
        off, nodes = self._gbwt.extract(np.asarray(ids, dtype=np.uint64), threads=2)
        self._last = _Rows(off, nodes)
        return self._last

    def extract_part_device(self, ids, part, parts):
        off, nodes = self._gbwt.extract(np.asarray(ids, dtype=np.uint64), threads=2)
        pieces, lens = [], []
        for k in range(len(off) - 1):
            row = nodes[int(off[k]):int(off[k + 1])]
            lo, hi = len(row) * part // parts, len(row) * (part + 1) // parts      # (the library cuts at sequence samples: any cut passes the same checks)
            pieces.append(row[lo:hi])
            lens.append(hi - lo)
        self._last = _Rows(np.concatenate([[0], np.cumsum(lens)]), np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.uint32))
        return self._last

    def last_kernel_ms(self):
        return 0.5, 0.6

    def last_offsets(self, n):
        return self._last.offsets[:n + 1]

    def path_sums(self, n):
        o, v = self._last.offsets, self._last.nodes.astype(np.uint64)
        return np.array([v[int(o[k]):int(o[k + 1])].sum(dtype=np.uint64) for k in range(n)], dtype=np.uint64)

    def path_hashes(self, n):
        return np.zeros(n, dtype=np.uint64)          # (compared at N = 1 only, by the cpu_baseline leg)

    def copy_path(self, k):
        o = self._last.offsets
        return self._last.nodes[int(o[k]):int(o[k + 1])]

    def sequences_csr(self, ids):
        r = self.extract_device(ids)
        return r.offsets, r.nodes

    # ---- GFA lines
    def path_lines_device(self, ids, mode):
        self._lines = _Lines([self._gbz.path_lines([int(p)], mode) for p in ids])
        return self._lines

    def last_lines_ms(self):
        return 0.2, 0.3

    def path_lines_array(self, ids, mode):
        return self.path_lines_device(ids, mode).text

    def path_lines(self, ids, mode):
        return self._gbz.path_lines([int(p) for p in ids], mode)


class FakeGBZ:
    @staticmethod
    def load(path, device=0, flags=None):
        if FAIL.startswith("c4_open:") and int(FAIL.split(":")[1]) == RANK and flags == G.OPEN_GFA:
            raise G.GbwtHipError(5, "rehearsal: this rank cannot open config 4's index")
        return FakeIndex(path)


class FakeGBWT:
    @staticmethod
    def from_records(data, starts, alphabet_offset, alphabet_size, sequences, size, bidirectional, device=0, flags=None):
        bwt = O.OracleBWT.from_parts(bytes(data), starts)
        f = FakeIndex.__new__(FakeIndex)
        f._gbz, f._gbwt, f._last, f._device = None, O.OracleGBWT.from_bwt(bwt, sequences, size, alphabet_offset, alphabet_size, bidirectional), None, 0
        return f


def main():
    # the handle classes -> the oracle; device tensors -> CPU tensors; "is there a GPU" -> yes (this IS the rehearsal of a GPU run)
    G.GBZ, G.GBWT = FakeGBZ, FakeGBWT
    D.paths_tensors = lambda paths, device: (torch.from_numpy(paths.offsets.astype(np.int64)), torch.from_numpy(paths.nodes.astype(np.int32)))
    D.lines_tensors = lambda lines, device: (torch.from_numpy(lines.offsets.copy()), torch.from_numpy(lines.text.copy()))
    real_device = torch.device
    torch.device = lambda *a, **k: real_device("cpu")
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    os.environ.setdefault("BENCH_DIST_BACKEND", "gloo")
    if FAIL.startswith("gather_hang:") and int(FAIL.split(":")[1]) == RANK:
        def hang(*a, **k):
            time.sleep(3600)
        D.gather_parts = hang
    if FAIL == "c4_generate" and RANK == 0:
        import c4_bench
        def broken(*a, **k):
            raise RuntimeError("rehearsal: the generator fails")
        c4_bench.generate = broken
    import bench
    t0 = time.perf_counter()
    bench.main()
    print(f"[rehearsal rank {RANK}] bench.main() returned after {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
