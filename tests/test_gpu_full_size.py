"""The BASELINE configs at the sizes they are quoted on, and the widths of the device side at their real limits (round 4).

Whatever the oracle finishes in seconds is compared with the oracle bit for bit (seeded samples); everything else with the
generator's ground truth and with the invariants of the domain.  Config 4's full size is in test_gpu_gfa.py, the headline's in
test_gpu_parity.py::test_headline_full_size."""
import os
import random
import sys

import numpy as np
import pytest

import gbwt_rs_amd as G
import oracle_lib as O
from gbwt_rs_amd import _lib
from gbwt_rs_amd import synth as S
from test_gpu_parity import make_queries, open_synth, oracle_of

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def states(st):
    return np.stack([st["node"], st["start"], st["end"]], axis=1)


def bd_states(bd):
    return np.stack([bd["forward"]["node"], bd["forward"]["start"], bd["forward"]["end"], bd["reverse"]["node"], bd["reverse"]["start"], bd["reverse"]["end"]], axis=1)


def test_config_c3_full_size():
    """BASELINE config 3 at its full size: bubble chain 1.1 M sites x 5 008 haplotypes (3.3 M nodes), a million queries of
    ten nodes built exactly as src/bin/benchmark.rs:124-153 builds them (start node uniform over the alphabet, offset uniform in its
    record, extended with GBWT::forward, discarded when the sequence ends early; seeded) -- the queries of bench.py's `search` object --
    plus a twentieth of them corrupted so that "not found" is exercised.  ALL of them go through the device, unidirectional
    (find + 9 x extend, src/bin/benchmark.rs:155-169) and bidirectional; every final state of a seeded sample of 30 000 is compared with
    the oracle, all of them with the invariants of a found state."""
    import configs as K
    keep = {}
    res = K.search(passes=1, keep=keep)
    dev, s, queries = keep["dev"], keep["synth"], keep["queries"]
    assert s.sites == 1100000 and s.paths == 5008 and len(queries) == 1000000 and res["unidirectional"]["found"] == len(queries)
    gen = np.random.default_rng(31)
    queries = queries.copy()
    bad = np.flatnonzero(gen.random(len(queries)) < 0.05)
    queries[bad, gen.integers(0, 10, len(bad))] ^= np.uint64(2)            # another node of the same orientation: usually no such path
    flip = gen.random(len(queries)) < 0.5                                  # half of them in the other orientation (benchmark.rs takes both strands' nodes as well)
    queries[flip] = (queries[flip] ^ np.uint64(1))[:, ::-1]
    queries = np.ascontiguousarray(queries)
    st, ok = dev.search(queries)
    bd, bok = dev.bd_search(queries, 4)
    assert 0.9 < ok.mean() < 1.0 and np.array_equal(ok, bok)               # found one way = found the other way
    assert np.array_equal(st["node"][ok], queries[ok][:, 9])
    assert np.array_equal((st["end"] - st["start"])[ok], (bd["forward"]["end"] - bd["forward"]["start"])[ok])
    assert np.array_equal((bd["forward"]["end"] - bd["forward"]["start"])[ok], (bd["reverse"]["end"] - bd["reverse"]["start"])[ok])
    assert np.array_equal(bd["reverse"]["node"][ok], queries[ok][:, 0] ^ np.uint64(1))
    # the same queries through the other ways in: a second workspace of the handle, and the device-resident forms (queries in HBM, states
    # left in the workspace)
    import torch
    plain = dev.another_workspace()
    st1, ok1 = plain.search(queries)
    bd1, bok1 = plain.bd_search(queries, 4)
    assert np.array_equal(ok, ok1) and np.array_equal(st, st1) and np.array_equal(bok, bok1) and np.array_equal(bd, bd1)
    d_q = torch.from_numpy(queries.view(np.int64)).cuda()
    st2, ok2 = plain.states_to_host(plain.search_device(d_q.data_ptr(), len(queries), 10))
    assert np.array_equal(ok, ok2) and np.array_equal(st, st2) and plain.last_query_ms() > 0
    bd2, bok2 = plain.states_to_host(plain.bd_search_device(d_q.data_ptr(), len(queries), 10, 4), bidirectional=True)
    assert np.array_equal(bok, bok2) and np.array_equal(bd, bd2)
    plain.close()
    del d_q
    # the same index opened for SEARCH only (gbwt_hip_open_records_flags): the same answers from a third of the memory
    everything = dev.memory_usage()["index_device_bytes"]
    lean = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, flags=G.OPEN_SEARCH)
    lean_bytes = lean.memory_usage()["index_device_bytes"]
    assert lean_bytes <= 4 << 30 and lean_bytes < 0.4 * everything, (lean_bytes, everything)
    st3, ok3 = lean.search(queries)
    bd3, bok3 = lean.bd_search(queries, 4)
    assert np.array_equal(st, st3) and np.array_equal(ok, ok3) and np.array_equal(bd, bd3) and np.array_equal(bok, bok3)
    lean.close()
    oracle = oracle_of(s)
    pick = np.sort(np.random.default_rng(8).choice(len(queries), 30000, replace=False))
    o_st, o_ok = oracle.search_batch(queries[pick], threads=16)
    assert np.array_equal(ok[pick], o_ok) and not o_ok.all()
    assert np.array_equal(states(st)[pick][o_ok], o_st[o_ok])
    o_bd, o_bok = oracle.bd_search_batch(queries[pick], 4, threads=16)
    assert np.array_equal(bok[pick], o_bok)
    assert np.array_equal(bd_states(bd)[pick][o_bok], o_bd[o_bok])
    dev.close()


def test_config_c5_full_size():
    """BASELINE config 5 at its full size: star chain of 3 000 sites with 300 alleles each, 5 000 haplotypes, Zipf(1.2)
    i.i.d. -- every anchor a record of outdegree 261 ... 288 (two-varint runs, ~10 KB), walked on the deep walk tables.  Every
    extracted path against the generator (per-path checksums on the device), 256 full rows of a seeded sample -- forward and
    reverse -- and 3 000 searches against the oracle."""
    s = S.Synth.chain(sites=3000, haplotypes=5000, alleles=300, model=S.IID, seed=42)
    dev = open_synth(s)
    assert dev.stats.max_outdegree >= 255
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    out = dev.extract_device(ids)
    assert int(out.total) == (s.size - s.sequences) // 2 == 5000 * 2 * 3000
    truth = np.array([s.path_checksum(h) for h in range(s.paths)], dtype=np.uint64)
    assert np.array_equal(dev.path_sums(s.paths), truth)
    oracle = oracle_of(s)
    rng = np.random.default_rng(5)
    sample = np.sort(rng.choice(s.sequences, 256, replace=False)).astype(np.uint64)
    offsets, nodes = dev.sequences_csr(sample)
    o_off, o_nodes = oracle.extract(sample, threads=16)
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    for k, seq in enumerate(sample[:8]):
        row, want = nodes[offsets[k]:offsets[k + 1]], s.path(int(seq) // 2)
        assert np.array_equal(row, want if seq % 2 == 0 else (want ^ 1)[::-1])
    queries = make_queries(s, random.Random(8), 3000, 6)
    st, ok = dev.search(queries)
    o_st, o_ok = oracle.search_batch(queries, threads=16)
    assert np.array_equal(ok, o_ok) and 0.3 < ok.mean() < 1.0
    assert np.array_equal(states(st)[ok], o_st[o_ok])
    bd, bok = dev.bd_search(queries, 2)
    o_bd, o_bok = oracle.bd_search_batch(queries, 2, threads=16)
    assert np.array_equal(bok, o_bok) and np.array_equal(bd_states(bd)[bok], o_bd[o_bok])
    m = dev.memory_usage()
    assert m["index_device_bytes"] > 16 * (s.size // 2)       # the tables are there: at least an LF-table entry per position of a table record
    dev.close()


def test_rank_blocks_beyond_4_gib():
    """An index whose packed half-blocks really exceed 4 GiB -- 13 000 haplotypes x 333 334 sites: 8.7 G positions in records of
    outdegree 2 = 135 M rank blocks of 32 bytes -- walked WITHOUT GBWT_HIP_WIDE_ADDRESSES: the loops must take 64-bit block addresses
    by themselves (walk_direct.hip: `narrow`).  Every path against the generator; rows from the far end of the block array in full."""
    assert "GBWT_HIP_WIDE_ADDRESSES" not in os.environ
    s = S.Synth.chain(sites=333334, haplotypes=13000, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=43)
    dev = open_synth(s)
    m = dev.memory_usage()
    assert m["index_device_bytes"] > (6 << 30), m                 # blocks (2.1 GB) + packed half-blocks (4.3 GB) + descriptors
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    out = dev.extract_device(ids)
    assert int(out.total) == 13000 * 2 * 333334
    truth = np.array([s.path_checksum(h) for h in range(s.paths)], dtype=np.uint64)
    assert np.array_equal(dev.path_sums(s.paths), truth)
    for h in (0, 6500, 12999):
        assert np.array_equal(dev.copy_path(h), s.path(h))
    rev = np.array([2 * 12999 + 1, 1], dtype=np.uint64)            # reverse sequences start at the far end of the chain: the last records first
    dev.extract_device(rev)
    assert np.array_equal(dev.copy_path(0), (s.path(12999) ^ 1)[::-1]) and np.array_equal(dev.copy_path(1), (s.path(0) ^ 1)[::-1])
    # search lands on the same blocks through block_follow: a few queries cut out of the last sites
    tail = s.path(77)[-40:].astype(np.uint64)
    queries = np.stack([tail[k:k + 10] for k in range(0, 30, 3)])
    st, ok = dev.search(queries)
    assert ok.all() and np.array_equal(st["node"], queries[:, 9]) and (st["end"] > st["start"]).all()
    dev.close()


def test_records_with_two_million_real_positions():
    """Records with 2^21 and more REAL positions (not GBWT_HIP_GATHER_LIMIT): 2.2 M haplotypes over a chain of three sites.  The packed
    half-blocks hold counts of 21 / 22 bits, so these records -- and the records behind their edges -- must be walked on the
    full-width blocks (`_full` loops), and `find` / `extend` return ranges beyond 2^21.  All paths against the generator, a seeded
    sample of sequences and searches against the oracle."""
    s = S.Synth.chain(sites=3, haplotypes=2200000, alleles=2, model=S.MOSAIC, founders=2, switch_rate=0.3, seed=12)
    dev = open_synth(s)
    assert dev.stats.max_record_len >= (1 << 21)
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    out = dev.extract_device(ids)
    assert int(out.total) == 2200000 * 6
    truth = np.array([s.path_checksum(h) for h in range(0, s.paths, 997)], dtype=np.uint64)
    assert np.array_equal(dev.path_sums(s.paths)[::997], truth)
    offsets, nodes = dev.sequences_csr(ids)
    assert np.all(np.diff(offsets) == 6)
    rows = nodes.reshape(-1, 6)
    for h in (0, 1, 1234567, 2199999):
        assert np.array_equal(rows[h], s.path(h))
    assert np.array_equal(rows[:, 0], np.full(len(rows), rows[0, 0])) and len(np.unique(rows[:, 1])) == 2      # one anchor, two alleles
    oracle = oracle_of(s)
    sample = np.sort(np.random.default_rng(3).choice(s.sequences, 2000, replace=False)).astype(np.uint64)
    g_off, g_nodes = dev.sequences_csr(sample)
    o_off, o_nodes = oracle.extract(sample, threads=16)
    assert np.array_equal(g_off, o_off) and np.array_equal(g_nodes, o_nodes)
    # search: the state of the first anchor covers all 2.2 M sequences; every prefix of a few paths against the oracle
    first = int(rows[0, 0])
    st, ok = dev.find([first])
    assert ok[0] and int(st["end"][0] - st["start"][0]) == 2200000 == oracle.find(first)[2] - oracle.find(first)[1]
    queries = np.ascontiguousarray(rows[::275000][:, :5].astype(np.uint64))
    st, ok = dev.search(queries)
    o_st, o_ok = oracle.search_batch(queries, threads=8)
    assert ok.all() and np.array_equal(ok, o_ok) and np.array_equal(states(st), o_st)
    assert (st["end"] - st["start"]).max() > (1 << 16)
    bd, bok = dev.bd_search(queries, 2)
    o_bd, o_bok = oracle.bd_search_batch(queries, 2, threads=8)
    assert np.array_equal(bok, o_bok) and np.array_equal(bd_states(bd)[bok], o_bd[o_bok])
    dev.close()


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def test_widths_beyond_u32_are_refused_at_open():
    """include/gbwt_hip.h "Widths": an index outside the 32-bit range of the device side is refused at open with GBWT_HIP_UNSUPPORTED --
    an alphabet above 2^32, and a (crafted, 20-byte) record with 2^32 positions: one run of sigma = 1, the byte 255 plus a varint
    (RLE::sanitize, src/support.rs:1292-1296: threshold 256)."""
    with pytest.raises(G.GbwtHipError) as e:
        G.GBWT.from_records(np.array([1, 2, 0, 0], dtype=np.uint8), [0], 0, (1 << 32) + 1, 1, 2, False)
    assert e.value.status == _lib.UNSUPPORTED and "alphabet_size" in str(e.value)
    # record 0 (endmarker): sigma 1, edge (node 1, offset 0), one run of one; record 1: sigma 1, edge (node 0 = ENDMARKER, 0), a run of 2^32
    rec0 = varint(1) + varint(1) + varint(0) + bytes([0])
    rec1 = varint(1) + varint(0) + varint(0) + bytes([255]) + varint((1 << 32) - 256)
    data = np.frombuffer(rec0 + rec1, dtype=np.uint8)
    with pytest.raises(G.GbwtHipError) as e:
        G.GBWT.from_records(data, [0, len(rec0)], 0, 2, 1, (1 << 32) + 1, False)
    assert e.value.status == _lib.UNSUPPORTED and "2^32" in str(e.value)
    # one position fewer is an index like any other (its one sequence has one node; positions past it belong to no sequence)
    rec1 = varint(1) + varint(0) + varint(0) + bytes([255]) + varint((1 << 32) - 257)
    data = np.frombuffer(rec0 + rec1, dtype=np.uint8)
    dev = G.GBWT.from_records(data, [0, len(rec0)], 0, 2, 1, 1 << 32, False)
    assert dev.stats.max_record_len == (1 << 32) - 1
    assert dev.sequence(0) == [1]
    st, ok = dev.find([1])
    assert ok[0] and int(st["end"][0]) == (1 << 32) - 1
    dev.close()


@pytest.mark.parametrize("after", [0, 2, 5])
def test_vmm_policy_after_field(monkeypatch, after):
    """GBWT_HIP_VMM = <chunk MiB>:<spread>:<min MiB>:<after>: a rows buffer of at least <min> that has served <after> requests as one
    hipMalloc is rebuilt from spread physical chunks (capi_internal.hpp: DeviceBuffer::reserve); after = 0 spreads at once.  The fourth
    field was not parsed until round 4.  Results are the same before and after the rebuild, and the memory comes back."""
    monkeypatch.setenv("GBWT_HIP_VMM", f"64:2:64:{after}")
    s = S.Synth.chain(sites=20000, haplotypes=600, alleles=2, model=S.MOSAIC, founders=8, switch_rate=0.01, seed=4)
    dev = open_synth(s)                                         # rows: 600 x 40 000 x 4 B = 96 MB
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    truth = np.array([s.path_checksum(h) for h in range(s.paths)], dtype=np.uint64)
    free_before = G.device_memory(0)[0]
    for request in range(1, after + 3):
        out = dev.extract_device(ids)
        assert int(out.total) == 600 * 40000 and np.array_equal(dev.path_sums(s.paths), truth)
        chunks = dev.memory_usage()["rows_chunks"]
        assert (chunks > 0) == (request > after), (request, after, chunks)
    assert dev.memory_usage()["rows_bytes"] >= 96000000
    dev.new_workspace()
    assert free_before - G.device_memory(0)[0] < (32 << 20), "the rows did not come back with their workspace"
    dev.close()
