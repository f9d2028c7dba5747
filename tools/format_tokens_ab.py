#!/usr/bin/env python3
"""Config 4 (size from argv, default small): the GFA formatter's two ways of putting a token together (GBWT_HIP_FORMAT_TOKENS, read per request: 0 = a
store per character, 1 = whole tokens from registers), interleaved in one process; the text of mode 1 is compared with mode 0's on the device."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4_bench
import gbwt_rs_amd as G
import torch
from gbwt_rs_amd import dist as D

size = sys.argv[1] if len(sys.argv) > 1 else "small"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
path = "/dev/shm/gbwt_c4_tokens.gbz"
g = c4_bench.generate(size, path)
generic = np.load(path + ".generic.npy")
walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
gbz = G.GBZ.load(path, flags=G.OPEN_GFA)
steps = (gbz.len() - gbz.sequences()) // 2


dev = torch.device("cuda", 0)


def text_of(lines):
    return D.lines_tensors(lines, dev)[1]


def request(mode, which, p_lines):
    os.environ["GBWT_HIP_FORMAT_TOKENS"] = str(mode)
    t0 = time.perf_counter()
    out = gbz.path_lines_device(which, p_lines)
    wall = (time.perf_counter() - t0) * 1e3
    walk, fmt = gbz.last_lines_ms()
    return out, wall, walk, fmt


# the first request of the paths fills the line cache
request(0, generic, 0)
request(0, walks, 1)
reference = {}
for which, p_lines, name in ((generic, 0, "P"), (walks, 1, "W")):
    out, *_ = request(0, which, p_lines)
    reference[name] = text_of(out).clone()
    gbz.path_lines_device(walks[:1], 1)
for mode in (1,):
    for which, p_lines, name in ((generic, 0, "P"), (walks, 1, "W")):
        out, *_ = request(mode, which, p_lines)
        assert c4_bench.device_bytes_equal(text_of(out), reference[name]), (mode, name)
        gbz.path_lines_device(walks[:1], 1)
print(f"text of mode 1 equal to mode 0's ({sum(int(v.numel()) for v in reference.values())} bytes)", flush=True)
reference.clear()
rows = {0: [], 1: []}
for _ in range(rounds):
    for mode in (0, 1):
        p = request(mode, generic, 0)
        w = request(mode, walks, 1)
        rows[mode].append((p[1] + w[1], p[2] + w[2], p[3] + w[3]))
        gbz.path_lines_device(walks[:1], 1)
for mode in (0, 1):
    wall, walk, fmt = (float(np.median([r[k] for r in rows[mode]])) for k in range(3))
    print(f"GBWT_HIP_FORMAT_TOKENS={mode}  wall {wall:8.3f} ms  walk kernel {walk:8.3f} ms  format stream {fmt:8.3f} ms  ({steps / wall / 1e6:6.1f} G LF-steps/s)", flush=True)
gbz.close()
c4_bench.cleanup(path)
