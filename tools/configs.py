#!/usr/bin/env python3
"""The BASELINE configs next to the headline, each as one function that returns the object bench.py puts on its line
(N = 1 only) and as a command `python3 tools/configs.py NAME` for the rocprofv3 passes of tools/measure_round.sh:

  secondary    the headline's shape with a one-node insertion as allele 1 of every site (rows leave lock step at once)
  high_degree  BASELINE config 5: 300 alleles per site (outdegree >= 255: two-varint runs), walked on the deep walk tables
  search       BASELINE config 3: 1.1 M sites x 5 008 haplotypes, 1 M queries of 10 nodes as src/bin/benchmark.rs:124-169 builds
               them (seeded), find + 9 x extend in one launch and the bidirectional form
  config4      BASELINE config 4 on one GPU at its stated size: ~42 000 walks over ~90 M nodes with labels of 1..1024 bp (tools/c4_bench.py)
  config4_small  the stand-in of rounds 3-4 (32 286 walks over 16 M one-base nodes), with the whole file written

Every object carries {workload, value, unit, kernel, kernel_ms, algorithmic_bytes}; bench.py adds the roofline fraction and the
measured HBM traffic of a PMC profile taken with the same sources and knobs (profiles/*_hbm_traffic.json, key = the config's name).
Results are checked against the generator's ground truth (extraction) or against invariants (search); parity with the oracle is
tests/'s business."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _timed_passes(index, ids, passes):
    walk, total, out = [], [], None
    for _ in range(passes):
        out = index.extract_device(ids)
        w, t = index.last_kernel_ms()
        walk.append(w)
        total.append(t)
    return out, walk, total


def _extraction(s, passes, warm, device, cpu_leg=None):
    import gbwt_rs_amd as G
    import torch
    dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, device=device)
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    _timed_passes(dev, ids, warm)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out, walk, _ = _timed_passes(dev, ids, passes)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as pool:      # (the generator's walk of every path: C code that runs without the interpreter lock)
        truth = np.array(list(pool.map(s.path_checksum, range(s.paths))), dtype=np.uint64)
    assert np.array_equal(dev.path_sums(len(ids)), truth), "extracted paths differ from the generator's ground truth"
    steps = int(out.total)
    res = {"value": steps * passes / elapsed, "unit": "LF-steps/s", "kernel": "k_walk_direct", "kernel_ms": float(np.mean(walk)),
           "value_kernel": steps / (float(np.mean(walk)) * 1e-3), "lf_steps": steps, "algorithmic_bytes": 4.0 * steps,
           "open_ms": dev.open_times()["total_ms"], "memory": dev.memory_usage()}
    if cpu_leg is not None:       # bench.py's cpu_baseline leg: the oracle over a bounded sample of these paths, compared with the rows just extracted
        res["cpu_baseline"] = cpu_leg(s, np.diff(dev.last_offsets(len(ids))), dev.path_sums(len(ids)), dev.path_hashes(len(ids)))
    dev.close()
    return res


def secondary(sites=333334, haplotypes=5000, model=0, seed=42, passes=5, device=0, cpu_leg=None):
    from gbwt_rs_amd import synth as S
    t0 = time.perf_counter()
    s = S.Synth.chain(sites=sites, haplotypes=haplotypes, alleles=2, model=model, founders=32, switch_rate=2e-3, seed=seed, extra=1)
    res = _extraction(s, passes, 3, device, cpu_leg)
    res["workload"] = (f"the headline's bubble chain with a one-node insertion as allele 1 of every site ({res['lf_steps']} LF-steps; rows of a batch "
                       "leave lock step after the first site: every wave is mixed)")
    res["seconds_incl_generator"] = round(time.perf_counter() - t0, 1)
    return res


def high_degree(haplotypes=5000, seed=42, passes=10, device=0, cpu_leg=None):
    from gbwt_rs_amd import synth as S
    t0 = time.perf_counter()
    s = S.Synth.chain(sites=3000, haplotypes=haplotypes, alleles=300, model=S.IID, seed=seed)
    res = _extraction(s, passes, 3, device, cpu_leg)
    res["workload"] = (f"BASELINE config 5: {haplotypes} haplotypes x 3 000 sites with 300 alleles each, i.i.d. Zipf(1.2) ({res['lf_steps']} LF-steps; "
                       "every site is a table record of outdegree >= 255, two-varint runs)")
    res["seconds_incl_generator"] = round(time.perf_counter() - t0, 1)
    return res


def make_benchmark_queries(dev, first, alphabet, n_queries, length, seed):
    """Queries as src/bin/benchmark.rs:124-153 builds them: a start node uniform in [first_node, alphabet_size), an offset uniform in its
    record, extended with GBWT::forward; discarded when the sequence ends early -- with a seeded generator (the reference's is not)."""
    import gbwt_rs_amd as G
    rng = np.random.default_rng(seed)
    want = int(n_queries * 1.3) + 1024
    nodes = rng.integers(first, alphabet, size=want, dtype=np.uint64)
    states, ok = dev.find(nodes)
    lens = (states["end"] - states["start"]).astype(np.uint64)
    keep = ok & (lens > 0)
    nodes, lens = nodes[keep], lens[keep]
    offsets = (rng.random(nodes.size) * lens).astype(np.uint64)
    pos = np.zeros(nodes.size, dtype=G.POS_DTYPE)
    pos["node"], pos["offset"] = nodes, offsets
    rows = [pos["node"].copy()]
    alive = np.ones(nodes.size, dtype=bool)
    for _ in range(length - 1):
        pos, ok = dev.forward(pos)
        alive &= ok & (pos["node"] != 0)
        rows.append(pos["node"].copy())
    return np.stack(rows, axis=1)[alive][:n_queries].astype(np.uint64)


def search(sites=1100000, haplotypes=5008, n_queries=1000000, length=10, seed=7, passes=5, device=0, keep=None, cpu_leg=None):
    import gbwt_rs_amd as G
    from gbwt_rs_amd import synth as S
    t0 = time.perf_counter()
    s = S.Synth.chain(sites, haplotypes, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=42)
    dev = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, device=device)
    queries = make_benchmark_queries(dev, s.alphabet_offset + 1, s.alphabet_size, n_queries, length, seed)
    n = queries.shape[0]
    res = {"workload": f"BASELINE config 3: bubble chain {sites} sites x {haplotypes} haplotypes (mosaic, seed 42), {n} queries of {length} nodes built as "
                       f"src/bin/benchmark.rs:124-153 does (seed {seed}); find + {length - 1} x extend per query in one launch, and bd_find + alternating "
                       "extend_forward / extend_backward",
           "queries": int(n), "unit": "queries/s"}
    # the kernel alone: queries resident in HBM, states left in the workspace (gbwt_hip_search_device); and the call as src/bin/benchmark.rs:161-164
    # times it: host pointers in and out (gbwt_hip_search: one pageable copy each way over the workspace stream), into FRESH result arrays and
    # into result arrays the caller has used before (profiles/r06_download_probe.txt: the kernel faults fresh pages in under the copy)
    import torch
    d_q = torch.from_numpy(queries.view(np.int64)).cuda(device)
    forms = (("unidirectional", "k_search", G.STATE_DTYPE, lambda w, o: w.search(queries, out=o), lambda: dev.states_to_host(dev.search_device(d_q.data_ptr(), n, length))),
             ("bidirectional", "k_bd_search", G.BD_DTYPE, lambda w, o: w.bd_search(queries, length // 2, out=o),
              lambda: dev.states_to_host(dev.bd_search_device(d_q.data_ptr(), n, length, length // 2), bidirectional=True)))
    for name, kernel, dtype, host_form, device_form in forms:
        kept = (np.zeros(n, dtype=dtype), np.zeros(n, dtype=np.uint8))
        host_form(dev, None), host_form(dev, kept), device_form()
        ks, ws, ws_kept = [], [], []
        for _ in range(passes):
            t1 = time.perf_counter()
            out, ok = host_form(dev, None)
            ws.append((time.perf_counter() - t1) * 1e3)
            t1 = time.perf_counter()
            out1, ok1 = host_form(dev, kept)
            ws_kept.append((time.perf_counter() - t1) * 1e3)
            out2, ok2 = device_form()
            ks.append(dev.last_query_ms())
        assert ok.all(), "a query cut out of the index itself was not found"
        assert np.array_equal(out, out2) and ok2.all() and np.array_equal(out, out1) and ok1.all(), "the ways into the kernel disagree"
        fwd = out if name == "unidirectional" else out["forward"]
        assert ((fwd["end"] > fwd["start"]) & (fwd["node"] == queries[:, -1 if name == "unidirectional" else length - 1])).all()
        k = float(np.mean(ks))
        # what a query must move in this layout: its nodes in (8 B each), its state out (24 / 48 B + 1), and per step the 64-byte descriptor of
        # the record + two 16-byte rank blocks (range start and end)
        bytes_q = 8 * length + (24 if name == "unidirectional" else 48) + 1 + length * (64 + 2 * 16)
        pcie = n * (8 * length + (24 if name == "unidirectional" else 48) + 1)
        res[name] = {"kernel": kernel, "kernel_ms": k, "wall_ms": float(np.median(ws)), "wall_ms_reused_results": float(np.median(ws_kept)), "value": n / (k * 1e-3),
                     "value_call": n / (float(np.median(ws)) * 1e-3), "value_call_reused_results": n / (float(np.median(ws_kept)) * 1e-3), "steps_per_s": n * length / (k * 1e-3), "ns_per_node_call": float(np.median(ws)) * 1e6 / (n * length),
                     "pcie_bytes_per_call": int(pcie), "pcie_GB_per_s": pcie / (float(np.median(ws)) * 1e-3) / 1e9,
                     "algorithmic_bytes": float(bytes_q * n), "found": int(ok.sum())}
        if name == "unidirectional":
            final_states, final_ok = out, ok
    del d_q
    res["value_call"] = res["unidirectional"]["value_call"]
    res["value_note"] = ("value = queries / kernel time with the queries resident in HBM (gbwt_hip_search_device); value_call = queries / wall time of "
                         "gbwt_hip_search with host pointers (src/bin/benchmark.rs:161-164 times the whole call): 80 MB in and 25 MB out over PCIe")
    if cpu_leg is not None:
        res["cpu_baseline"] = cpu_leg(s, queries, final_states, final_ok)
    res["value"] = res["unidirectional"]["value"]
    res["kernel"], res["kernel_ms"], res["algorithmic_bytes"] = "k_search", res["unidirectional"]["kernel_ms"], res["unidirectional"]["algorithmic_bytes"]
    res["memory"] = dev.memory_usage()
    res["seconds_incl_generator"] = round(time.perf_counter() - t0, 1)
    if keep is not None:
        keep.update(dev=dev, synth=s, queries=queries)
    else:
        dev.close()
    return res


def config4(passes=3, out="", device=0, size="full", cpu_leg=None):
    """BASELINE config 4 on one GPU at the size SURVEY 8(d) states (tools/c4_bench.py: SIZES["full"]); size="small" is the stand-in of rounds
    3-4 (bench.py's `config4_small`, with the whole file written to /dev/shm)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import c4_bench
    try:
        res = c4_bench.run(size=size, passes=passes, out=out, device=device, cpu_leg=cpu_leg)
    finally:
        if out and os.path.exists(out):
            os.remove(out)
    wf = res["walk_format"]
    res.update({"value": wf["value"], "unit": "LF-steps/s", "kernel": "k_walk_direct + k_format_chunks", "kernel_ms": wf["ms"],
                "value_first_request": wf["value_first_request"], "first_request_ms": wf["first_request_ms"],
                "first_request_device_ms": wf["first_request_device_ms"], "first_request_host_ms": wf["first_request_host_ms"],
                "algorithmic_bytes": float(wf["bytes_moved"])})
    return res


def config4_small(passes=5, out="/dev/shm/gbwt_bench_c4.gfa", device=0, cpu_leg=None):
    return config4(passes, out, device, size="small", cpu_leg=cpu_leg)


if __name__ == "__main__":
    # usage: configs.py NAME [NAME ...] [--device D] [--c4-size SIZE] [--cpu-leg SECONDS]
    # one name: its object as one JSON line (tools/measure_round.sh); several: {name: object}.  --cpu-leg: with bench.py's CPU leg of the config
    # (the oracle over a bounded sample, results compared) -- bench.py starts config 4 this way, in a process of its own.
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+", choices=["secondary", "high_degree", "search", "config4", "config4_small"])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--c4-size", default="full")
    ap.add_argument("--cpu-leg", type=float, default=0.0)
    a = ap.parse_args()
    legs = {}
    if a.cpu_leg > 0:
        import bench
        legs = {"secondary": bench.cpu_leg_extraction(a.cpu_leg), "high_degree": bench.cpu_leg_extraction(a.cpu_leg), "search": bench.cpu_leg_search(a.cpu_leg),
                "config4": bench.cpu_leg_lines(a.cpu_leg), "config4_small": bench.cpu_leg_lines(a.cpu_leg)}
    out = {}
    for name in a.names:
        if name == "config4":
            out[name] = config4(device=a.device, size=a.c4_size, cpu_leg=legs.get(name))
        else:
            fn = {"secondary": secondary, "high_degree": high_degree, "search": search, "config4_small": config4_small}[name]
            out[name] = fn(device=a.device, cpu_leg=legs.get(name))
    print(json.dumps(out[a.names[0]] if len(a.names) == 1 else out), flush=True)
