"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference's known answers.

Bit-exact for everything (integer/index work).  Mirrors the reference's own tests:
src/bwt/tests.rs:159-237 (lf / follow over every offset and range), src/gbwt/tests.rs:164-462
(extract, sequence, find, extend, bidirectional search), src/gbz/tests.rs:85-98,371-381 (GBZ::path).
"""
import os
import random

import numpy as np
import pytest

import gbwt_rs_amd as G
import kat
import oracle_lib as O
from gbwt_rs_amd import synth as S

pytestmark = pytest.mark.gpu

GOLDEN = O.GOLDEN


def states(rows):
    return np.array(rows, dtype=G.STATE_DTYPE)


def bd_states(rows):
    return np.array(rows, dtype=G.BD_DTYPE)


def bd_tuple(x):
    return (tuple(int(v) for v in x["forward"]), tuple(int(v) for v in x["reverse"]))


def open_synth(s):
    return G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, s.bidirectional)


def oracle_of(s):
    bwt = O.OracleBWT.from_parts(bytes(s.data()), s.starts())
    return O.OracleGBWT.from_bwt(bwt, s.sequences, s.size, s.alphabet_offset, s.alphabet_size, s.bidirectional)


def check_all_positions(dev, oracle):
    """start + forward for every position of every record, find/extend/bd for every state the oracle can reach."""
    n_seq = oracle.sequences()
    pos, ok = dev.start(np.arange(n_seq + 2))
    for i in range(n_seq + 2):
        exp = oracle.start(i)
        assert bool(ok[i]) == (exp is not None)
        if exp is not None:
            assert tuple(int(v) for v in pos[i]) == exp
    # every (node, offset) incl. one past the end of each record and nodes outside the alphabet
    queries = []
    for node in range(0, oracle.alphabet_size() + 2):
        st = oracle.find(node)
        ln = (st[2] - st[1]) if st else 0
        for off in range(ln + 2):
            queries.append((node, off))
    out, ok = dev.forward(np.array(queries, dtype=G.POS_DTYPE))
    for q, r, v in zip(queries, out, ok):
        exp = oracle.forward(q)
        assert bool(v) == (exp is not None), q
        if exp is not None:
            assert tuple(int(x) for x in r) == exp, q
    if oracle.is_bidirectional():
        # GBWT::backward for the same positions (src/gbwt/tests.rs:191-214 walks every sequence backward)
        out, ok = dev.backward(np.array(queries, dtype=G.POS_DTYPE))
        for q, r, v in zip(queries, out, ok):
            exp = oracle.backward(q)
            assert bool(v) == (exp is not None), q
            if exp is not None:
                assert tuple(int(x) for x in r) == exp, q


def check_search(dev, oracle, nodes_of_interest):
    nodes = np.arange(0, oracle.alphabet_size() + 2, dtype=np.uint64)
    st, ok = dev.find(nodes)
    for node, s, v in zip(nodes, st, ok):
        exp = oracle.find(int(node))
        assert bool(v) == (exp is not None)
        if exp:
            assert tuple(int(x) for x in s) == exp
    # extend: every sub-range of every record x every destination
    q_states, q_nodes = [], []
    for node in nodes_of_interest:
        f = oracle.find(node)
        if f is None:
            continue
        ln = f[2]
        ranges = [(a, b) for a in range(ln + 1) for b in range(a, ln + 2)] if ln <= 6 else \
                 [(0, ln), (0, 1), (ln - 1, ln), (1, ln - 1), (ln // 2, ln // 2), (ln, ln + 3)]
        for (a, b) in ranges:
            for dest in range(0, oracle.alphabet_size() + 2):
                q_states.append((node, a, b))
                q_nodes.append(dest)
    out, ok = dev.extend(states(q_states), q_nodes)
    for s, d, r, v in zip(q_states, q_nodes, out, ok):
        exp = oracle.extend(s, d)
        assert bool(v) == (exp is not None), (s, d)
        if exp:
            assert tuple(int(x) for x in r) == exp, (s, d)
    if not oracle.is_bidirectional():
        return
    bst, ok = dev.bd_find(nodes)
    for node, s, v in zip(nodes, bst, ok):
        exp = oracle.bd_find(int(node))
        assert bool(v) == (exp is not None)
        if exp:
            assert bd_tuple(s) == exp
    # bidirectional extension: forward range from the extend cases, arbitrary reverse range of the same length
    q_bd = []
    for (node, a, b) in q_states[::2]:
        for rev_start in (0, 2):
            q_bd.append(((node, a, b), (node ^ 1, rev_start, rev_start + max(0, b - a))))
    dests = list(range(0, oracle.alphabet_size() + 2))
    rng = random.Random(7)
    q_nodes = [rng.choice(dests) for _ in q_bd]
    for fn, ofn in ((dev.extend_forward, oracle.extend_forward), (dev.extend_backward, oracle.extend_backward)):
        out, ok = fn(bd_states(q_bd), q_nodes)
        for s, d, r, v in zip(q_bd, q_nodes, out, ok):
            exp = ofn(s, d)
            assert bool(v) == (exp is not None), (s, d)
            if exp:
                assert bd_tuple(r) == exp, (s, d)


def check_follow(dev, oracle, limit=400):
    """GBZ::follow_forward / follow_backward the way src/gbz/tests.rs:100-168 exercises them: breadth-first over every
    state reachable from every node, device extensions against the oracle's, in edge order."""
    if not oracle.is_bidirectional():
        return
    nodes = list(range(0, oracle.alphabet_size() + 2))
    frontier, seen = [], set()
    for node in nodes:
        st = oracle.bd_find(node)
        if st is not None and st not in seen:
            seen.add(st)
            frontier.append(st)
    # states for nodes that do not exist must give "no iterator"
    bogus = [((n, 0, 1), (n ^ 1, 0, 1)) for n in (0, 1, oracle.alphabet_size(), oracle.alphabet_size() + 1)]
    checked = 0
    while frontier and checked < limit:
        batch, frontier = frontier[:128], frontier[128:]
        queries = batch + (bogus if checked == 0 else [])
        checked += len(batch)
        for backward in (False, True):
            offsets, ext, ok = dev.follow(bd_states(queries), backward)
            for k, st in enumerate(queries):
                exp = oracle.follow(st, backward)
                assert bool(ok[k]) == (exp is not None), (st, backward)
                got = [bd_tuple(e) for e in ext[int(offsets[k]):int(offsets[k + 1])]]
                assert got == (exp or []), (st, backward)
                for e in exp or []:
                    if e not in seen:
                        seen.add(e)
                        frontier.append(e)


# ---------------------------------------------------------------------------------------------
# reference fixtures


@pytest.mark.parametrize("name,with_empty", [("example.gbwt", False), ("with-empty.gbwt", True)])
def test_fixture_extract(name, with_empty):
    """src/gbwt/tests.rs:164-189, 216-238: all sequences vs the known paths, both orientations."""
    dev = G.GBWT.load(os.path.join(GOLDEN, name))
    truth = kat.true_paths(with_empty)
    assert dev.sequences() == 2 * len(truth)
    offsets, nodes = dev.sequences_csr(np.arange(dev.sequences()))
    assert int(offsets[-1]) == dev.len() - dev.sequences()
    for i, t in enumerate(truth):
        assert list(nodes[offsets[2 * i]:offsets[2 * i + 1]]) == t
        assert list(nodes[offsets[2 * i + 1]:offsets[2 * i + 2]]) == kat.reverse_path(t)
        assert dev.sequence(2 * i) == t
    assert dev.sequence(dev.sequences()) is None
    # GBWT::sequence(id >= sequences) is None (src/gbwt/tests.rs:227): a value, not an error -- the id gets an empty row
    # and a False in the mask, the rest of the batch is extracted
    mixed = np.array([dev.sequences(), 0, 2 ** 40, 3, dev.sequences() + 1], dtype=np.uint64)
    m_off, m_nodes, m_valid = dev.sequences_csr(mixed, return_valid=True)
    assert list(m_valid) == [False, True, False, True, False]
    assert list(np.diff(m_off.astype(np.int64))) == [0, len(truth[0]), 0, len(truth[1]), 0]
    assert list(m_nodes) == truth[0] + kat.reverse_path(truth[1])
    if with_empty:
        assert dev.sequence(8) == [] and dev.sequence(9) == []
    # the per-row checksums of the device-resident rows (gbwt_hip_path_sums / gbwt_hip_path_hashes) against the oracle's walk of the same ids
    dev.extract_device(mixed)
    steps, o_lens, o_sums, o_hashes = O.OracleGBWT.load(os.path.join(GOLDEN, name)).extract_checksums(mixed, 2)
    assert steps == len(m_nodes) and np.array_equal(np.diff(dev.last_offsets(len(mixed))), o_lens)
    assert np.array_equal(dev.path_sums(len(mixed)), o_sums) and np.array_equal(dev.path_hashes(len(mixed)), o_hashes)


def test_size_then_fill_computes_once():
    """The C idiom -- size query, then the same call with buffers -- must not walk the sequences twice: the second call
    finds the rows of the same request in the workspace.  A different request in between is computed afresh."""
    import ctypes as C
    dev = G.GBWT.load(os.path.join(GOLDEN, "example.gbwt"))
    L = dev._L
    truth = kat.true_paths(False)
    ids = np.array([0, 2, 6, 99, 1], dtype=np.uint64)
    offsets = np.zeros(ids.size + 1, dtype=np.uint64)
    total = C.c_uint64(0)
    G._lib.check(L.gbwt_hip_extract(dev._h, dev._ws, ids.ctypes.data, ids.size, offsets.ctypes.data, None, 0, C.byref(total)))
    walk_ms = dev.last_kernel_ms()
    nodes = np.zeros(total.value, dtype=np.uint32)
    G._lib.check(L.gbwt_hip_extract(dev._h, dev._ws, ids.ctypes.data, ids.size, offsets.ctypes.data, nodes.ctypes.data, nodes.size, C.byref(total)))
    assert dev.last_kernel_ms() == walk_ms            # same events: no second walk
    want = truth[0] + truth[1] + truth[3] + kat.reverse_path(truth[0])
    assert list(nodes) == want and list(np.diff(offsets.astype(np.int64))) == [5, 4, 5, 0, 5]
    # another request, then the first one again with buffers only
    other = np.array([4], dtype=np.uint64)
    o2 = np.zeros(2, dtype=np.uint64)
    n2 = np.zeros(16, dtype=np.uint32)
    G._lib.check(L.gbwt_hip_extract(dev._h, dev._ws, other.ctypes.data, 1, o2.ctypes.data, n2.ctypes.data, n2.size, C.byref(total)))
    assert list(n2[:total.value]) == truth[2]
    nodes[:] = 0
    G._lib.check(L.gbwt_hip_extract(dev._h, dev._ws, ids.ctypes.data, ids.size, offsets.ctypes.data, nodes.ctypes.data, nodes.size, C.byref(total)))
    assert list(nodes) == want
    # too small a buffer: CAPACITY, and the total is still reported
    st = L.gbwt_hip_extract(dev._h, dev._ws, ids.ctypes.data, ids.size, offsets.ctypes.data, nodes.ctypes.data, 3, C.byref(total))
    assert st == G._lib.CAPACITY and total.value == len(want)
    # GBZ::path through the C entry point (sequence id = 2 * path + orientation)
    pids = np.array([1, 0], dtype=np.uint64)
    po = np.zeros(3, dtype=np.uint64)
    pn = np.zeros(16, dtype=np.uint32)
    G._lib.check(L.gbwt_hip_extract_paths(dev._h, dev._ws, pids.ctypes.data, 2, 1, po.ctypes.data, pn.ctypes.data, pn.size, C.byref(total)))
    assert list(pn[:total.value]) == kat.reverse_path(truth[1]) + kat.reverse_path(truth[0])
    # follow: size query + fill
    st8, ok = dev.bd_find([28, 24])
    off, ext, valid = dev.follow(st8)
    off2, ext2, valid2 = dev.follow(st8)
    assert ok.all() and valid.all() and np.array_equal(off, off2) and np.array_equal(ext, ext2) and len(ext) == int(off[-1]) > 0
    # follow: a size query, then another query on the same workspace (it reuses the staging buffers), then the fill call
    st6 = np.ascontiguousarray(st8, dtype=G.BD_DTYPE)
    f_off = np.zeros(st6.size + 1, dtype=np.uint64)
    f_valid = np.zeros(st6.size, dtype=np.uint8)
    G._lib.check(L.gbwt_hip_follow(dev._h, dev._ws, st6.ctypes.data, st6.size, 0, f_off.ctypes.data, None, 0, C.byref(total), f_valid.ctypes.data))
    dev.find([22, 42, 30])
    f_out = np.zeros(total.value, dtype=G.BD_DTYPE)
    G._lib.check(L.gbwt_hip_follow(dev._h, dev._ws, st6.ctypes.data, st6.size, 0, f_off.ctypes.data, f_out.ctypes.data, f_out.size, C.byref(total), f_valid.ctypes.data))
    assert np.array_equal(f_off, off) and np.array_equal(f_out, ext)
    # gbwt_hip_copy_result: offsets only, nodes only, too small a buffer
    p = dev.extract_device(ids)
    only_off = np.zeros(ids.size + 1, dtype=np.uint64)
    G._lib.check(L.gbwt_hip_copy_result(dev._h, dev._ws, only_off.ctypes.data, None, 0))
    assert list(np.diff(only_off.astype(np.int64))) == [5, 4, 5, 0, 5] and int(only_off[-1]) == p.total
    only_nodes = np.zeros(p.total, dtype=np.uint32)
    G._lib.check(L.gbwt_hip_copy_result(dev._h, dev._ws, None, only_nodes.ctypes.data, only_nodes.size))
    assert list(only_nodes) == want
    assert L.gbwt_hip_copy_result(dev._h, dev._ws, None, only_nodes.ctypes.data, 3) == G._lib.CAPACITY
    assert L.gbwt_hip_copy_result(dev._h, dev._ws, None, None, 0) == G._lib.BAD_ARGUMENT
    fresh = G.GBWT.load(os.path.join(GOLDEN, "example.gbwt"))
    assert L.gbwt_hip_copy_result(fresh._h, fresh._ws, only_off.ctypes.data, None, 0) == G._lib.BAD_ARGUMENT   # nothing extracted yet


def test_fixture_statistics():
    dev = G.GBWT.load(os.path.join(GOLDEN, "example.gbwt"))
    assert (dev.len(), dev.sequences(), dev.alphabet_size(), dev.alphabet_offset()) == (68, 12, 52, 21)
    assert dev.is_bidirectional() and dev.first_node() == 22 and dev.has_metadata()
    assert dev.stats.max_record_len == 12 and dev.stats.max_outdegree == 4


@pytest.mark.parametrize("name", ["example.gbwt", "with-empty.gbwt"])
def test_fixture_navigation_and_search(name):
    path = os.path.join(GOLDEN, name)
    dev, oracle = G.GBWT.load(path), O.OracleGBWT.load(path)
    check_all_positions(dev, oracle)
    check_search(dev, oracle, sorted(kat.true_nodes()))
    check_follow(dev, oracle)


def test_fixture_search_known_answers():
    """doc-test src/gbwt.rs:70-83 and the brute-force counts of src/gbwt/tests.rs:294-350 / 393-462."""
    dev = G.GBWT.load(os.path.join(GOLDEN, "example.gbwt"))
    st, ok = dev.search([[24, 28, 30]])
    assert ok[0] and int(st[0]["node"]) == 30 and int(st[0]["end"] - st[0]["start"]) == 2
    bd, ok = dev.bd_find([28])
    bd, ok = dev.extend_backward(bd, [24])
    bd, ok = dev.extend_forward(bd, [30])
    assert ok[0] and bd_tuple(bd[0]) == ((30, 0, 2), (25, 0, 2))
    paths = kat.true_paths(False)
    queries, expected = [], []
    for p in paths:
        for p2 in (p, kat.reverse_path(p)):
            for j in range(len(p2)):
                for k in range(j + 1, len(p2) + 1):
                    queries.append(p2[j:k])
                    expected.append(kat.count_occurrences(paths, p2[j:k]))
    for ln in sorted({len(q) for q in queries}):
        qs = [q for q in queries if len(q) == ln]
        ex = [e for q, e in zip(queries, expected) if len(q) == ln]
        st, ok = dev.search(qs)
        assert ok.all()
        assert [int(s["end"] - s["start"]) for s in st] == ex
    st, ok = dev.search([[24, 30], [0, 22], [22, 0], [60, 22]])
    assert not ok.any()


@pytest.mark.parametrize("name", ["example.gbz", "example-v1.gbz", "translation.gbz", "translation-v1.gbz"])
def test_fixture_gbz_paths(name):
    """src/gbz/tests.rs:85-98, 279-292, 371-381: GBZ::path in both orientations."""
    path = os.path.join(GOLDEN, name)
    dev, oracle = G.GBZ.load(path), O.OracleGBZ(path)
    truth = kat.true_paths(False) if name.startswith("example") else [[2 * x for x in p] for p in kat.TRANSLATION_PATHS]
    assert dev.paths() == len(truth) == oracle.paths()
    for i, t in enumerate(truth):
        assert dev.path(i) == [(x // 2, x & 1) for x in t] == oracle.path(i)
        assert dev.path(i, G.REVERSE) == [(x // 2, x & 1) for x in kat.reverse_path(t)]
    assert dev.path(dev.paths()) is None
    offsets, nodes = dev.paths_csr(np.arange(dev.paths()))
    o_off, o_nodes = oracle.gbwt().extract(np.arange(0, 2 * dev.paths(), 2))
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)


def test_unidirectional_index_rejects_bd_calls():
    """Where the reference asserts (src/gbwt.rs:312,340), the C ABI returns an error status."""
    bwt = O.OracleBWT(kat.PAPER_EDGES, kat.PAPER_RUNS)
    dev = G.GBWT.from_records(bwt.data(), bwt.starts(), 0, 8, 3, 17, bidirectional=False)
    with pytest.raises(G.GbwtHipError):
        dev.bd_find([1])
    with pytest.raises(G.GbwtHipError):
        dev.backward(np.array([(1, 0)], dtype=G.POS_DTYPE))
    st, ok = dev.find([1, 2, 7, 8, 0])
    assert list(ok) == [True, True, True, False, False]


# ---------------------------------------------------------------------------------------------
# paper examples through gbwt_hip_open_records (src/bwt/tests.rs:10-87)


@pytest.mark.parametrize("edges,runs,n_seq,bidirectional", [(kat.PAPER_EDGES, kat.PAPER_RUNS, 3, False),
                                                            (kat.BD_EDGES, kat.BD_RUNS, 6, True)])
def test_paper_examples(edges, runs, n_seq, bidirectional):
    bwt = O.OracleBWT(edges, runs)
    size = sum(l for r in runs for _, l in r)
    data, starts = bwt.data(), bwt.starts()
    # unidirectional example: node v <-> record v; bidirectional: GBWT nodes 2..15 <-> records 1..14 (offset 1)
    offset = 1 if bidirectional else 0
    alphabet = len(edges) + offset
    dev = G.GBWT.from_records(data, starts, offset, alphabet, n_seq, size, bidirectional=bidirectional)
    oracle = O.OracleGBWT.from_bwt(O.OracleBWT.from_parts(data, starts), n_seq, size, offset, alphabet, bidirectional)
    check_all_positions(dev, oracle)
    check_search(dev, oracle, list(range(offset + 1, alphabet)))
    check_follow(dev, oracle)
    offsets, nodes = dev.sequences_csr(np.arange(n_seq))
    o_off, o_nodes = oracle.extract(np.arange(n_seq))
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)


def test_empty_records():
    """src/bwt/tests.rs:314-329: records 2 and 6 are the single byte 0x00."""
    edges = [list(e) for e in kat.PAPER_EDGES]
    runs = [list(r) for r in kat.PAPER_RUNS]
    for k in (2, 6):
        edges[k], runs[k] = [], []
    bwt = O.OracleBWT(edges, runs)
    dev = G.GBWT.from_records(bwt.data(), bwt.starts(), 0, 8, 3, 17, bidirectional=False)
    oracle = O.OracleGBWT.from_bwt(O.OracleBWT.from_parts(bwt.data(), bwt.starts()), 3, 17, 0, 8, False)
    check_all_positions(dev, oracle)
    st, ok = dev.find([2, 6])
    assert not ok.any()


# ---------------------------------------------------------------------------------------------
# generated indexes


@pytest.mark.parametrize("seed,cyclic", [(11, False), (12, True), (13, True)])
def test_random_path_sets(seed, cyclic):
    rng = random.Random(seed)
    paths = []
    for _ in range(40):
        ln = rng.randint(0, 30)
        if cyclic:
            paths.append([2 * rng.randint(1, 12) + rng.randint(0, 1) for _ in range(ln)])
        else:
            paths.append([2 * i for i in sorted(rng.sample(range(1, 41), ln))])
    paths[0] = [2, 4, 6]
    s = S.Synth.from_paths(paths, bidirectional=True)
    dev, oracle = open_synth(s), oracle_of(s)
    ids = list(range(s.sequences)) + [3, 3, 0]   # duplicates are allowed
    offsets, nodes = dev.sequences_csr(ids)
    for k, i in enumerate(ids):
        exp = paths[i // 2] if i % 2 == 0 else kat.reverse_path(paths[i // 2])
        assert list(nodes[offsets[k]:offsets[k + 1]]) == exp
    check_all_positions(dev, oracle)
    nodes_used = sorted({x for p in paths for x in p} | {x ^ 1 for p in paths for x in p})
    check_search(dev, oracle, nodes_used[:24])
    check_follow(dev, oracle)



def _check_positions_sampled(dev, oracle, rng, per_record=24):
    """forward / backward at the first, the last, one past the last and random offsets of every record; find + extend + bd over
    whole records and random sub-ranges."""
    queries, q_states, q_nodes = [], [], []
    first = oracle.alphabet_offset() + 1
    for node in range(first, oracle.alphabet_size()):
        st = oracle.find(node)
        ln = st[2] if st else 0
        offs = {0, ln, ln + 1, max(0, ln - 1)} | {rng.randrange(0, ln + 1) for _ in range(per_record)}
        queries += [(node, o) for o in sorted(offs)]
        if st:
            a = rng.randrange(0, ln); b = rng.randrange(a, ln + 1)
            for rng_ in ((0, ln), (a, b)):
                for _ in range(3):
                    fw = oracle.forward((node, rng.randrange(rng_[0], max(rng_[0] + 1, rng_[1]))))
                    q_states.append((node, rng_[0], rng_[1])); q_nodes.append(fw[0] if fw else first)
    out, ok = dev.forward(np.array(queries, dtype=G.POS_DTYPE))
    bout, bok = dev.backward(np.array(queries, dtype=G.POS_DTYPE))
    for q, r, v, br, bv in zip(queries, out, ok, bout, bok):
        exp, bexp = oracle.forward(q), oracle.backward(q)
        assert bool(v) == (exp is not None) and (exp is None or tuple(int(x) for x in r) == exp), q
        assert bool(bv) == (bexp is not None) and (bexp is None or tuple(int(x) for x in br) == bexp), q
    out, ok = dev.extend(states(q_states), q_nodes)
    for st, d, r, v in zip(q_states, q_nodes, out, ok):
        exp = oracle.extend(st, d)
        assert bool(v) == (exp is not None) and (exp is None or tuple(int(x) for x in r) == exp), (st, d)
    q_bd = [((n, a, b), (n ^ 1, 0, b - a)) for (n, a, b) in q_states]
    for fn, ofn in ((dev.extend_forward, oracle.extend_forward), (dev.extend_backward, oracle.extend_backward)):
        out, ok = fn(bd_states(q_bd), q_nodes)
        for st, d, r, v in zip(q_bd, q_nodes, out, ok):
            exp = ofn(st, d)
            assert bool(v) == (exp is not None) and (exp is None or bd_tuple(r) == exp), (st, d)


@pytest.mark.parametrize("alleles,haplotypes", [(150, 2500), (253, 4000), (128, 2500)])
def test_rle_one_byte_plus_varint_regime(alleles, haplotypes):
    """129 <= sigma <= 254: the run-length threshold is 1, so EVERY run is one byte (the value) + a varint (length - 1)
    (src/support.rs:1292-1296, 1413-1430; src/support/tests.rs:461-469) -- the regime between the byte-packed runs of small
    alphabets and the two-varint runs of sigma >= 255, and the one that stresses RunDecoder's reciprocal division."""
    s = S.Synth.chain(sites=10, haplotypes=haplotypes, alleles=alleles, model=S.IID, zipf=0.0, seed=alleles)
    dev, oracle = open_synth(s), oracle_of(s)
    assert 128 < dev.stats.max_outdegree < 255, dev.stats.max_outdegree
    ids = np.arange(s.sequences, dtype=np.uint64)
    offsets, nodes = dev.sequences_csr(ids)
    o_off, o_nodes = oracle.extract(ids, threads=8)
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    # every offset of the widest records (the anchors), sampled offsets elsewhere
    rng = random.Random(alleles)
    first = s.alphabet_offset + 1
    wide = sorted(range(first, s.alphabet_size), key=lambda n: -((oracle.find(n) or (0, 0, 0))[2]))[:3]
    queries = [(n, o) for n in wide for o in range(oracle.find(n)[2] + 2)]
    out, ok = dev.forward(np.array(queries, dtype=G.POS_DTYPE))
    for q, r, v in zip(queries, out, ok):
        exp = oracle.forward(q)
        assert bool(v) == (exp is not None) and (exp is None or tuple(int(x) for x in r) == exp), q
    _check_positions_sampled(dev, oracle, rng, per_record=4)
    for mode in (1, 2):                                      # the lane-serial and the wave-cooperative decoder read the same streams
        dev.tune(walk_mode=mode)
        offsets, nodes = dev.sequences_csr(ids[:400])
        assert np.array_equal(nodes, o_nodes[:o_off[400]])


def test_long_runs_in_small_alphabets():
    """sigma = 1: threshold 256, a run of 256 or more is the byte 255 + varint(length - 256) -- here with three-byte varints
    (runs of more than 2^14 + 256); sigma = 2: threshold 128 (src/support.rs:1292-1296; src/support/tests.rs:439-459)."""
    trunk = [2 * v for v in range(1, 7)]
    paths = [trunk] * 21000 + [trunk[:3] + [2 * 9] + trunk[4:]] * 400 + [trunk] * 18000 + [[2 * 9, 2 * 3 + 1]] * 3
    s = S.Synth.from_paths(paths, bidirectional=True)
    dev, oracle = open_synth(s), oracle_of(s)
    assert dev.stats.max_record_len >= (1 << 14) + 256
    run = 21000 + 400 + 18000 - 256                          # node 1's record: one run of 39 400 -> 255, then varint(39 144) in three bytes
    assert bytes([255, 0x80 | (run & 0x7F), 0x80 | ((run >> 7) & 0x7F), run >> 14]) in bytes(s.data())
    ids = np.arange(s.sequences, dtype=np.uint64)
    offsets, nodes = dev.sequences_csr(ids)
    o_off, o_nodes = oracle.extract(ids, threads=8)
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    _check_positions_sampled(dev, oracle, random.Random(5), per_record=64)
    for mode in (1, 2, 3):
        dev.tune(walk_mode=mode)
        offsets, nodes = dev.sequences_csr(ids[::97])
        e_off, e_nodes = oracle.extract(ids[::97], threads=8)
        assert np.array_equal(offsets, e_off) and np.array_equal(nodes, e_nodes)

@pytest.mark.parametrize("alleles,model,zipf", [(2, S.MOSAIC, 1.2), (2, S.IID, 1.2), (7, S.IID, 1.0), (300, S.IID, 0.2), (400, S.IID, 0.0)])
def test_chain_indexes(alleles, model, zipf):
    """Bubble / star chains incl. the sigma >= 255 two-varint regime and runs longer than one byte can hold."""
    s = S.Synth.chain(sites=40, haplotypes=1500 if alleles > 100 else 700, alleles=alleles, model=model, founders=6,
                      switch_rate=0.05, zipf=zipf, seed=alleles)
    dev, oracle = open_synth(s), oracle_of(s)
    if alleles >= 300:
        assert dev.stats.max_outdegree >= 255
    ids = np.arange(s.sequences, dtype=np.uint64)
    offsets, nodes = dev.sequences_csr(ids)
    o_off, o_nodes = oracle.extract(ids, threads=4)
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    for h in (0, 1, s.paths - 1):
        assert np.array_equal(nodes[offsets[2 * h]:offsets[2 * h + 1]], s.path(h))
    # random positions / searches against the oracle
    rng = random.Random(alleles)
    first = s.alphabet_offset + 1
    qs = []
    for _ in range(300):
        node = rng.randrange(first, s.alphabet_size)
        f = oracle.find(node)
        qs.append((node, rng.randrange(0, (f[2] if f else 1) + 1)))
    out, ok = dev.forward(np.array(qs, dtype=G.POS_DTYPE))
    for q, r, v in zip(qs, out, ok):
        exp = oracle.forward(q)
        assert bool(v) == (exp is not None) and (exp is None or tuple(int(x) for x in r) == exp)
    # queries of length 6 cut out of true paths (all must be found) + corrupted ones
    queries = []
    for _ in range(200):
        p = s.path(rng.randrange(s.paths))
        a = rng.randrange(0, len(p) - 6)
        q = [int(x) for x in p[a:a + 6]]
        if rng.random() < 0.3:
            q[rng.randrange(6)] ^= 2
        if rng.random() < 0.3:
            q = kat.reverse_path(q)
        queries.append(q)
    st, ok = dev.search(queries)
    for q, r, v in zip(queries, st, ok):
        exp = oracle.find(q[0])
        for x in q[1:]:
            exp = oracle.extend(exp, x) if exp else None
        assert bool(v) == (exp is not None), q
        if exp:
            assert tuple(int(x) for x in r) == exp
    # bidirectional walk: bd_find(q[2]) -> forward over q[3:], backward over q[1], q[0]
    mid, ok = dev.bd_find([q[2] for q in queries])
    exp = [oracle.bd_find(q[2]) for q in queries]
    cur, cur_ok = mid, ok
    for step, (fn, ofn, col) in enumerate([(dev.extend_forward, oracle.extend_forward, 3), (dev.extend_forward, oracle.extend_forward, 4),
                                           (dev.extend_backward, oracle.extend_backward, 1), (dev.extend_backward, oracle.extend_backward, 0)]):
        nodes_col = [q[col] for q in queries]
        nxt, nxt_ok = fn(cur, nodes_col)
        for k in range(len(queries)):
            e = ofn(exp[k], nodes_col[k]) if (exp[k] is not None) else None
            if exp[k] is not None:
                assert bool(nxt_ok[k]) == (e is not None), (queries[k], step)
                if e:
                    assert bd_tuple(nxt[k]) == e
            exp[k] = e
            if e is None:
                nxt[k] = bd_states([((0, 0, 0), (0, 0, 0))])[0]
        cur, cur_ok = nxt, nxt_ok


def test_config_c2_bit_exact(tmp_path):
    """BASELINE config 2: synthetic 1k paths x 10k nodes (3,333 sites, seed 42), both models, via a .gbz file."""
    for model in (S.MOSAIC, S.IID):
        s = S.Synth.chain(sites=3333, haplotypes=1000, alleles=2, model=model, founders=32, switch_rate=2e-3, seed=42)
        path = tmp_path / f"c2-{model}.gbz"
        s.save(str(path), as_gbz=True)
        dev = G.GBZ.load(str(path))
        oracle = O.OracleGBZ(str(path)).gbwt()
        ids = np.arange(0, 2000, 2, dtype=np.uint64)       # forward sequences = what gbunzip extracts
        offsets, nodes = dev.sequences_csr(ids)
        o_off, o_nodes = oracle.extract(ids, threads=8)
        assert int(offsets[-1]) == 1000 * 6666
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
        ids = np.arange(1, 2000, 2, dtype=np.uint64)       # reverse sequences
        offsets, nodes = dev.sequences_csr(ids)
        o_off, o_nodes = oracle.extract(ids, threads=8)
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)


@pytest.mark.parametrize("alleles,extra,model", [(2, 1, S.MOSAIC), (2, 3, S.IID), (4, 2, S.MOSAIC)])
def test_indel_chain_bit_exact(tmp_path, alleles, extra, model):
    """Insertion alleles: paths of different lengths whose walks leave lock step after the first site (the wave-uniform
    loop hardly ever applies).  Forward and reverse sequences against the oracle, via a .gbz file."""
    s = S.Synth.chain(sites=2500, haplotypes=600, alleles=alleles, model=model, founders=16, switch_rate=5e-3, zipf=0.7, seed=31, extra=extra)
    path = tmp_path / "indel.gbz"
    s.save(str(path), as_gbz=True)
    dev = G.GBZ.load(str(path))
    oracle = O.OracleGBZ(str(path)).gbwt()
    for first in (0, 1):
        ids = np.arange(first, 1200, 2, dtype=np.uint64)
        offsets, nodes = dev.sequences_csr(ids)
        o_off, o_nodes = oracle.extract(ids, threads=8)
        assert len(set(np.diff(offsets).tolist())) > 1
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    assert dev.path_lines(np.arange(1, 40, dtype=np.uint64), 1) == O.OracleGBZ(str(path)).path_lines(list(range(1, 40)), 1)


@pytest.mark.parametrize("env", [{"GBWT_HIP_WIDE_ADDRESSES": "1"}, {"GBWT_HIP_RING_SLOTS": "32"}, {"GBWT_HIP_UNIFORM_LOOP": "0"},
                                 {"GBWT_HIP_SAMPLE_INTERVAL": "256", "GBWT_HIP_LOOKAHEAD_HOPS": "0"},
                                 {"GBWT_HIP_GATHER_LIMIT": "300"}, {"GBWT_HIP_GATHER_LIMIT": "0", "GBWT_HIP_WIDE_ADDRESSES": "1"},
                                 {"GBWT_HIP_CATCH_UP": "2"}, {"GBWT_HIP_CATCH_UP": "2", "GBWT_HIP_GATHER_LIMIT": "300"}, {"GBWT_HIP_CATCH_UP": "2", "GBWT_HIP_SAMPLE_INTERVAL": "64"}])
def test_walk_loop_variants(monkeypatch, env):
    """The loops of k_walk_direct with 64-bit addresses, with a ring asked for that is smaller than two row pieces (the library
    raises it: a 32-slot ring never holds a 128-byte piece and the walk would not end), as the only loop, with short
    segments, and with records declared too long for the packed counts of its blocks (some / all of them: such waves move
    to the loops on the full-width blocks); sparse and dense insertions, every path against the generator's allele matrix."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for extra, every in ((1, 1), (1, 37), (0, 1)):    # (0, 1): the plain chain -- the uniform loop, on packed or (GATHER_LIMIT) full-width blocks
        s = S.Synth.chain(sites=6000, haplotypes=700, alleles=2, model=S.MOSAIC, founders=16, switch_rate=5e-3, seed=77, extra=extra, indel_every=every)
        dev = open_synth(s)
        ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
        out = dev.extract_device(ids)
        assert int(out.total) == (s.size - s.sequences) // 2
        sums = dev.path_sums(len(ids))
        assert all(int(sums[h]) == s.path_checksum(h) for h in range(s.paths))
        for h in (0, 350, 699):
            assert np.array_equal(dev.copy_path(h), s.path(h))


@pytest.mark.parametrize("chains", ["6", "0", "1", "3", "6 pool"])
@pytest.mark.parametrize("chop,extra,every", [(3, 0, 1), (4, 1, 2), (2, 2, 1), (9, 5, 3), (1, 7, 1)])
def test_chopped_chain_bit_exact(tmp_path, monkeypatch, chop, extra, every, chains):
    """Every node a chain of `chop` nodes with consecutive ids -- most records unary, as in a GBZ built from a GFA with long
    segments -- with and without insertions: forward and reverse sequences and W-lines against the oracle.  GBWT_HIP_CHAINS: steps
    that run through up to that many more unary records with consecutive ids (k_link_desc2; 0 = fused pairs only, 6 = the default);
    chains longer than the limit are taken in several steps, reverse sequences walk them with descending ids."""
    monkeypatch.setenv("GBWT_HIP_CHAINS", chains.split()[0])
    if chains.endswith("pool"):
        monkeypatch.setenv("GBWT_HIP_DIRECT", "0")       # the pool-output kernel: chained steps in plain C++ (k_walk_two)
    s = S.Synth.chain(sites=1200, haplotypes=500, alleles=2, model=S.MOSAIC, founders=16, switch_rate=5e-3, seed=41, extra=extra, indel_every=every, chop=chop)
    path = tmp_path / "chopped.gbz"
    s.save(str(path), as_gbz=True)
    dev = G.GBZ.load(str(path))
    oracle = O.OracleGBZ(str(path)).gbwt()
    for first in (0, 1):
        ids = np.arange(first, 1000, 2, dtype=np.uint64)
        offsets, nodes = dev.sequences_csr(ids)
        o_off, o_nodes = oracle.extract(ids, threads=8)
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    assert dev.path_lines(np.arange(1, 30, dtype=np.uint64), 1) == O.OracleGBZ(str(path)).path_lines(list(range(1, 30)), 1)


def _segment_paths(segments, haplotypes, seed):
    """Paths over GFA-like segments chopped into 1 .. 12 nodes with consecutive ids: a path takes a segment forwards (ids ascending,
    orientation +) or backwards (ids descending, orientation -), skips some, ends inside some -- runs of unary records in both
    orientations, longer and shorter than a chained step can hold, with merges and branches in between."""
    rng = random.Random(seed)
    first, ids = [], 1
    for _ in range(segments):
        n = rng.choice([1, 1, 2, 3, 4, 7, 8, 9, 12])
        first.append((ids, n))
        ids += n + rng.choice([0, 0, 1])            # sometimes the next segment's ids run on, sometimes there is a gap
    paths = []
    for h in range(haplotypes):
        p = []
        for k, (f, n) in enumerate(first):
            r = rng.random()
            if r < 0.15:
                continue
            if r < 0.25:
                p.extend(2 * (f + n - 1 - i) + 1 for i in range(n))      # the segment backwards
            else:
                p.extend(2 * (f + i) for i in range(n))
        if h % 11 == 0 and p:
            p = p[:rng.randint(1, len(p))]
        paths.append(p)
    return [p for p in paths if p]


@pytest.mark.parametrize("bidirectional", [True, False], ids=["bidirectional", "unidirectional"])
@pytest.mark.parametrize("env", [{}, {"GBWT_HIP_CHAINS": "6"}, {"GBWT_HIP_CHAINS": "2", "GBWT_HIP_SAMPLE_INTERVAL": "8"}, {"GBWT_HIP_CHAINS": "6", "GBWT_HIP_SAMPLE_INTERVAL": "0"},
                                 {"GBWT_HIP_CHAINS": "5", "GBWT_HIP_UNIFORM_LOOP": "0", "GBWT_HIP_SAMPLE_INTERVAL": "17"},
                                 {"GBWT_HIP_CHAINS": "6", "GBWT_HIP_GATHER_LIMIT": "0", "GBWT_HIP_SAMPLE_INTERVAL": "33"}],
                         ids=lambda e: ",".join(f"{k[9:]}={v}" for k, v in e.items()) or "defaults")
def test_chained_steps_over_chopped_segments(monkeypatch, env, bidirectional):
    """Chained steps (k_link_desc2: a step runs through a run of unary records with consecutive ids, forwards with ascending and backwards with
    descending ids; never through a record with two predecessors in a bidirectional index, anywhere in a unidirectional one) against the oracle:
    every sequence, forward() from every position, find / extend over the same records (which know nothing of chains)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for seed, (segments, haplotypes) in enumerate([(40, 30), (300, 400), (25, 1200)]):
        s = S.Synth.from_paths(_segment_paths(segments, haplotypes, 70 + seed), bidirectional=bidirectional)
        dev, oracle = open_synth(s), oracle_of(s)
        ids = np.arange(0, s.sequences, dtype=np.uint64)
        o_off, o_nodes = oracle.extract(ids, threads=4)
        offsets, nodes = dev.sequences_csr(ids)
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes), (env, seed)
        if seed == 0:
            check_all_positions(dev, oracle)


@pytest.mark.parametrize("extra,every", [(1, 1), (1, 37), (3, 64), (0, 1)])
def test_lean_extract_handle_walks_like_the_full_one(extra, every):
    """A handle opened for extraction only (gbwt_hip_open_records_flags without SEARCH) on an index none of whose records needs the generic
    decoder gives back its raw descriptors, its one-step walk descriptors and its plain rank blocks (128 B per record + 16 B per 64
    positions); the catch-up steps of lagging lanes then read the two-step descriptors and packed half-blocks.  Dense and sparse insertions
    (catch-up at work) and the plain chain: every path against the generator, rows against the full handle, parts of rows too."""
    s = S.Synth.chain(sites=20000, haplotypes=1500, alleles=2, model=S.MOSAIC, founders=16, switch_rate=5e-3, seed=78, extra=extra, indel_every=every)
    full = open_synth(s)
    lean = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, flags=G.OPEN_EXTRACT)
    records = int(full.stats.records)
    assert lean.memory_usage()["index_device_bytes"] <= full.memory_usage()["index_device_bytes"] - 128 * records
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    out = lean.extract_device(ids)
    assert int(out.total) == (s.size - s.sequences) // 2
    sums = lean.path_sums(len(ids))
    assert all(int(sums[h]) == s.path_checksum(h) for h in range(s.paths))
    full.extract_device(ids)
    assert np.array_equal(lean.path_hashes(len(ids)), full.path_hashes(len(ids)))
    some = np.array([1, 2 * 700 + 1, 5, 2 * 1499], dtype=np.uint64)                 # reverse sequences too
    a, b = lean.sequences_csr(some), full.sequences_csr(some)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for part in range(3):
        x, y = lean.part_csr(ids[:200], part, 3), full.part_csr(ids[:200], part, 3)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])


@pytest.mark.parametrize("env", [{}, {"GBWT_HIP_FUSED_OFFSETS": "0"}, {"GBWT_HIP_SEGMENTS": "0"}, {"GBWT_HIP_SAMPLE_INTERVAL": "0"},
                                 {"GBWT_HIP_SAMPLE_INTERVAL": "16", "GBWT_HIP_SAMPLE_STRIDE": "3"}],
                         ids=lambda e: ",".join(f"{k[9:]}={v}" for k, v in e.items()) or "defaults")
def test_rows_of_one_length_need_no_offsets_launch(monkeypatch, env):
    """Where every sequence of the index has the same number of nodes (the headline's shape, config 5) a batch of valid ids knows its row
    offsets without a table: the walkers compute them and k_walk_direct writes the n + 1 offsets the caller reads (WalkArgs::uniform_len;
    GBWT_HIP_FUSED_OFFSETS=0: k_row_offsets in front of the walk, as before round 6).  Shuffled ids with repeats, both orientations, one row,
    7 000 rows, and a batch with an id that does not exist (not uniform: the table path) -- offsets and nodes against the oracle's walk."""
    s = S.Synth.chain(sites=700, haplotypes=160, alleles=2, model=S.MOSAIC, founders=8, switch_rate=1e-2, seed=21)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = open_synth(s)
    gbwt = oracle_of(s)
    rng = np.random.default_rng(3)
    batches = [rng.integers(0, s.sequences, size=500).astype(np.uint64), np.array([s.sequences - 1], dtype=np.uint64),
               rng.integers(0, s.sequences, size=7000).astype(np.uint64), np.array([3, s.sequences + 5, 0, 1], dtype=np.uint64)]
    for ids in batches:
        off, nodes = dev.sequences_csr(ids)
        o_off, o_nodes = gbwt.extract(ids, threads=4)
        assert np.array_equal(off, o_off) and np.array_equal(nodes, o_nodes), (env, len(ids))
        out = dev.extract_device(ids)
        assert np.array_equal(dev.last_offsets(len(ids)), o_off) and int(out.total) == int(o_off[-1])


@pytest.mark.parametrize("flags", ["OPEN_EXTRACT", "OPEN_GFA"])
def test_lean_handle_refuses_walks_without_samples(monkeypatch, flags):
    """A lean handle has no raw descriptors, and a walker that does not start from a sequence sample reads them when it arrives on its
    first record (walk_loops.hpp: arrive).  With GBWT_HIP_SEGMENTS=0 -- whole rows and, since such rows cannot be cut, the last of several
    parts -- the request is refused with GBWT_HIP_UNSUPPORTED before anything is launched (round 5 would have read through a null device
    pointer); earlier parts are n empty rows as the header says, rows of no nodes are served, and the same handle with a default
    workspace still walks like the full one."""
    s = S.Synth.chain(sites=3000, haplotypes=200, alleles=2, model=S.MOSAIC, founders=8, switch_rate=5e-3, seed=5)
    full = open_synth(s)
    lean = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, flags=getattr(G, flags))
    ids = np.arange(0, 64, 2, dtype=np.uint64)
    want = full.sequences_csr(ids)
    monkeypatch.setenv("GBWT_HIP_SEGMENTS", "0")
    lean.new_workspace()
    for call in (lambda: lean.sequences_csr(ids), lambda: lean.part_csr(ids, 2, 3)):
        with pytest.raises(G.GbwtHipError) as e:
            call()
        assert e.value.status == _lib_status("UNSUPPORTED")
    off, nodes = lean.part_csr(ids, 0, 3)
    assert off.tolist() == [0] * (len(ids) + 1) and len(nodes) == 0
    beyond = np.array([s.sequences, s.sequences + 7], dtype=np.uint64)          # GBWT::sequence -> None: empty rows, nothing to walk
    assert lean.sequences_csr(beyond)[0].tolist() == [0, 0, 0]
    monkeypatch.delenv("GBWT_HIP_SEGMENTS")
    lean.new_workspace()
    got = lean.sequences_csr(ids)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    for part in range(3):
        x, y = lean.part_csr(ids, part, 3), full.part_csr(ids, part, 3)
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])


@pytest.mark.parametrize("extra,every,sites", [(2, 1, 60000), (3, 64, 150000), (1, 8, 100000), (3, 4096, 150000)])
def test_indel_chain_scale_properties(extra, every, sites):
    """The same regime at a size the oracle does not finish quickly: every path against the generator's allele matrix.  Dense
    insertions (waves mixed for good: the gather loop), sparse ones (waves a step apart that catch up by single steps and return to the
    uniform loop, with the ring filling up on the way) and in between (attempts that do not pay and back off)."""
    s = S.Synth.chain(sites=sites, haplotypes=3000, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=9, extra=extra, indel_every=every)
    dev = open_synth(s)
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    out = dev.extract_device(ids)
    assert int(out.total) == (s.size - s.sequences) // 2
    sums = dev.path_sums(len(ids))
    for h in range(s.paths):
        assert int(sums[h]) == s.path_checksum(h)
    for h in (0, 1, 1500, 2999):
        assert np.array_equal(dev.copy_path(h), s.path(h))
    r_off, r_nodes = dev.sequences_csr([2 * 1500 + 1, 1])
    assert np.array_equal(r_nodes[r_off[0]:r_off[1]], (s.path(1500) ^ 1)[::-1])
    assert np.array_equal(r_nodes[r_off[1]:r_off[2]], (s.path(0) ^ 1)[::-1])


def test_headline_scale_properties():
    """Size-independent checks at a large size (paths x sites well beyond what the oracle finishes quickly):
    every extracted path equals the generator's ground truth, lengths are uniform, the total equals
    (size - sequences) / 2, and reverse extraction is the flipped reversal of forward extraction."""
    s = S.Synth.chain(sites=40000, haplotypes=2048, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=7)
    dev = open_synth(s)
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    offsets, nodes = dev.sequences_csr(ids)
    assert int(offsets[-1]) == (s.size - s.sequences) // 2
    assert np.all(np.diff(offsets) == 2 * s.sites)
    sums = np.add.reduceat(nodes.astype(np.uint64), offsets[:-1].astype(np.int64))
    for h in range(s.paths):
        assert int(sums[h]) == s.path_checksum(h)
    for h in (0, 1, 777, 2047):
        assert np.array_equal(nodes[offsets[h]:offsets[h + 1]], s.path(h))
    r_off, r_nodes = dev.sequences_csr([2 * 777 + 1, 1])
    assert np.array_equal(r_nodes[r_off[0]:r_off[1]], (s.path(777) ^ 1)[::-1])
    assert np.array_equal(r_nodes[r_off[1]:r_off[2]], (s.path(0) ^ 1)[::-1])


def test_headline_full_size():
    """BASELINE's headline index at its full size (5 000 paths x 1 000 002 nodes, 3.33 G LF-steps per pass): every extracted path
    against the generator's ground truth (per-path checksums reduced on the device, full rows for a few), EVERY path against the
    ORACLE's walk (length / sum / order-dependent hash of all 5 000 rows, whole rows for a seeded sample of 64), uniform lengths, total = (size - sequences) / 2,
    reverse sequences = flipped reversals, and the same answers from a second pass (the extraction is idempotent)."""
    s = S.Synth.chain(sites=333334, haplotypes=5000, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=42)
    dev = open_synth(s)
    ids = np.arange(0, s.sequences, 2, dtype=np.uint64)
    truth = np.array([s.path_checksum(h) for h in range(s.paths)], dtype=np.uint64)
    for _ in range(2):
        out = dev.extract_device(ids)
        assert int(out.total) == (s.size - s.sequences) // 2 == 5000 * 2 * 333334
        assert np.array_equal(dev.path_sums(s.paths), truth)
    for h in (0, 1234, 4999):
        row = dev.copy_path(h)
        assert len(row) == 2 * s.sites and np.array_equal(row, s.path(h))
    # ... and against the ORACLE at this size (every path, not a sample): ALL 5 000 forward sequences walked by the CPU
    # restatement of SequenceIter (src/gbwt.rs:557-568) -- 3.33 G LF-steps, about half a minute on 64 host threads -- with the length, node
    # sum and ORDER-DEPENDENT hash of every row compared (gbwt_hip_path_sums, gbwt_hip_path_hashes); whole rows for a seeded sample of 64
    oracle = oracle_of(s)
    threads = min(os.cpu_count() or 1, 64)
    o_steps, o_lens, o_sums, o_hashes = oracle.extract_checksums(ids, threads=threads)
    assert o_steps == 5000 * 2 * s.sites
    lens, sums, hashes = np.diff(dev.last_offsets(s.paths)), dev.path_sums(s.paths), dev.path_hashes(s.paths)
    assert np.array_equal(lens, o_lens) and np.array_equal(sums, o_sums) and np.array_equal(hashes, o_hashes), "a row differs from the oracle's walk"
    sample = np.sort(np.random.default_rng(64).choice(s.paths, size=64, replace=False)).astype(np.uint64)
    o_off, o_nodes = oracle.extract(2 * sample, threads=threads)
    assert int(o_off[-1]) == 64 * 2 * s.sites
    for k, h in enumerate(sample):
        assert np.array_equal(dev.copy_path(int(h)), o_nodes[int(o_off[k]):int(o_off[k + 1])]), f"path {h} differs from the oracle's walk"
    rev = np.array([2 * 1234 + 1, 1, 2 * 4999 + 1], dtype=np.uint64)
    dev.extract_device(rev)
    for k, h in enumerate((1234, 0, 4999)):
        assert np.array_equal(dev.copy_path(k), (s.path(h) ^ 1)[::-1])
    o_off, o_nodes = oracle.extract(rev, threads=3)
    assert np.array_equal(dev.path_hashes(3), oracle.extract_checksums(rev, 3)[3]) and np.array_equal(dev.copy_path(2), o_nodes[int(o_off[2]):])
    # the same batch as eight GPUs (and three) share it: stretch r of EVERY path (gbwt_hip_extract_part_device).  The stretches of a row
    # follow each other without gap or overlap, their checksums add up to the row's, and a few rows are put together and compared whole
    for parts in (8, 3):
        at, sums, steps = np.zeros(s.paths, dtype=np.uint64), np.zeros(s.paths, dtype=np.uint64), 0
        pieces = {h: [] for h in (0, 1234, 4999)}
        for r in range(parts):
            out = dev.extract_part_device(ids, r, parts)
            steps += int(out.total)
            lens = np.diff(dev.last_offsets(s.paths))
            assert lens.min() > 0 and lens.max() < 2 * (2 * s.sites // parts), (parts, r, int(lens.min()), int(lens.max()))   # (every rank gets its share of every row)
            sums += dev.path_sums(s.paths)                                   # (uint64: wraps like the generator's)
            for h in pieces:
                pieces[h].append(dev.copy_path(h))
            at += lens
        assert steps == 5000 * 2 * 333334 and np.all(at == 2 * s.sites) and np.array_equal(sums, truth), parts
        for h, got in pieces.items():
            assert np.array_equal(np.concatenate(got), s.path(h)), (parts, h)
    # the same index opened for EXTRACT only (gbwt_hip_open_records_flags): no raw / one-step descriptors, no plain rank blocks -- two thirds of
    # the memory, the same rows
    lean = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, flags=G.OPEN_EXTRACT)
    lean_bytes, all_bytes = lean.memory_usage()["index_device_bytes"], dev.memory_usage()["index_device_bytes"]
    assert lean_bytes <= 2.4e9 and lean_bytes < 0.72 * all_bytes, (lean_bytes, all_bytes)
    lean.extract_device(ids)
    assert np.array_equal(lean.path_sums(s.paths), truth)
    lean.close()
    # The 13.3 GB of rows start as one hipMalloc and are rebuilt from spread 2 GiB chunks (virtual-memory API) when the workspace
    # serves its third request of that size; every byte must come back when the workspace goes -- one hipMemUnmap per mapped chunk,
    # hipMemAddressFree, hipMemRelease (capi_internal.hpp: DeviceBuffer::release) -- and when it regrows.
    free_before = G.device_memory(0)[0]
    view = dev.another_workspace()
    for _ in range(4):                                     # the third one rebuilds the rows from spread chunks
        out = view.extract_device(ids)
    assert np.array_equal(view.path_sums(s.paths), truth)
    used = free_before - G.device_memory(0)[0]
    assert used >= 4 * int(out.total), used                # the rows are there ...
    bigger = np.concatenate([ids, ids[:600]])              # ... a larger batch regrows them (release + a new hipMalloc) ...
    out = view.extract_device(bigger)
    assert int(out.total) == 5600 * 2 * 333334
    assert free_before - G.device_memory(0)[0] < 4 * int(out.total) + (3 << 30), "the old rows were not given back when the buffer regrew"
    view.close()
    leaked = free_before - G.device_memory(0)[0]
    assert leaked < (256 << 20), f"{leaked} bytes of VRAM did not come back with the workspace"


# ---------------------------------------------------------------------------------------------
# kernel variants: results never depend on the tuning

TUNINGS = [dict(walk_mode=0, paths_per_wave=64, small_record=16),   # default: two LF steps per iteration on rank blocks
           dict(walk_mode=0, paths_per_wave=7, small_record=16),
           dict(walk_mode=2, paths_per_wave=64, small_record=16),   # cooperative for long records (bpermute search)
           dict(walk_mode=2, paths_per_wave=64, small_record=0),    # every class 1/2 record through the cooperative path
           dict(walk_mode=2, paths_per_wave=5, small_record=0),     # few owners per wave (member-by-member resolution)
           dict(walk_mode=2, paths_per_wave=17, small_record=40),
           dict(walk_mode=1, paths_per_wave=64, small_record=16),   # lane-serial kernel
           dict(walk_mode=3, paths_per_wave=64, small_record=16),   # one LF step per iteration on rank blocks
           dict(walk_mode=3, paths_per_wave=5, small_record=16)]


def variant_cases():
    yield "fixture", None
    yield "mosaic-long-runs", dict(sites=60, haplotypes=3000, alleles=2, model=S.MOSAIC, founders=2, switch_rate=0.01, seed=3)
    yield "iid-many-runs", dict(sites=30, haplotypes=3000, alleles=2, model=S.IID, seed=4)
    yield "single-allele-sites", dict(sites=50, haplotypes=700, alleles=2, model=S.MOSAIC, founders=1, switch_rate=0.0, seed=5)
    yield "multi-allelic", dict(sites=40, haplotypes=900, alleles=4, model=S.IID, zipf=0.5, seed=6)


@pytest.mark.parametrize("name,params", list(variant_cases()))
def test_walk_variants_agree_with_oracle(name, params):
    if params is None:
        path = os.path.join(GOLDEN, "with-empty.gbwt")
        dev, oracle = G.GBWT.load(path), O.OracleGBWT.load(path)
        n_seq = oracle.sequences()
    else:
        s = S.Synth.chain(**params)
        dev, oracle = open_synth(s), oracle_of(s)
        n_seq = s.sequences
    ids = np.arange(n_seq, dtype=np.uint64)
    o_off, o_nodes = oracle.extract(ids, threads=8)
    for t in TUNINGS:
        dev.tune(**t)
        offsets, nodes = dev.sequences_csr(ids)
        assert np.array_equal(offsets, o_off), (name, t)
        assert np.array_equal(nodes, o_nodes), (name, t)
    # ragged batch: a few sequences, repeated ids, odd count
    rng = random.Random(1)
    some = [rng.randrange(n_seq) for _ in range(37)]
    o_off, o_nodes = oracle.extract(some)
    for t in TUNINGS:
        dev.tune(**t)
        offsets, nodes = dev.sequences_csr(some)
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes), (name, t)


def test_random_paths_all_variants():
    rng = random.Random(99)
    paths = [[2 * rng.randint(1, 6) + rng.randint(0, 1) for _ in range(rng.randint(0, 300))] for _ in range(150)]
    s = S.Synth.from_paths(paths, bidirectional=True)
    dev = open_synth(s)
    ids = np.arange(s.sequences, dtype=np.uint64)
    for t in TUNINGS:
        dev.tune(**t)
        offsets, nodes = dev.sequences_csr(ids)
        for i, p in enumerate(paths):
            assert list(nodes[offsets[2 * i]:offsets[2 * i + 1]]) == p, t
            assert list(nodes[offsets[2 * i + 1]:offsets[2 * i + 2]]) == kat.reverse_path(p), t


# ---------------------------------------------------------------------------------------------
# config 3 shape: batched unidirectional and bidirectional search at scale


def make_queries(s, rng, n_queries, length, corrupt=0.2):
    """Queries like src/bin/benchmark.rs:124-153 (subpaths of real paths, forward or reverse), some corrupted."""
    queries = np.zeros((n_queries, length), dtype=np.uint64)
    for k in range(n_queries):
        p = s.path(rng.randrange(s.paths))
        a = rng.randrange(0, len(p) - length)
        q = p[a:a + length].astype(np.uint64)
        if rng.random() < 0.5:
            q = (q ^ 1)[::-1]
        if rng.random() < corrupt:
            q = q.copy()
            q[rng.randrange(length)] ^= 2
        queries[k] = q
    return queries


@pytest.mark.parametrize("model,haplotypes", [(S.MOSAIC, 5008), (S.IID, 1000)])
def test_config_c3_search_bit_exact(model, haplotypes):
    """BASELINE config 3 (chr22-scale stand-in, reduced number of sites): find + 9 x extend and the bidirectional
    walk for 20 000 queries of 10 nodes, every final state compared with the oracle."""
    s = S.Synth.chain(sites=3000, haplotypes=haplotypes, alleles=2, model=model, founders=32, switch_rate=2e-3, seed=22)
    dev, oracle = open_synth(s), oracle_of(s)
    rng = random.Random(3)
    queries = make_queries(s, rng, 20000, 10)
    st, ok = dev.search(queries)
    o_st, o_ok = oracle.search_batch(queries, threads=8)
    assert np.array_equal(ok, o_ok) and 0.5 < ok.mean() < 1.0
    got = np.stack([st["node"], st["start"], st["end"]], axis=1)
    assert np.array_equal(got[ok], o_st[o_ok])
    for first in (0, 4, 9):
        bd, bok = dev.bd_search(queries, first)
        o_bd, o_bok = oracle.bd_search_batch(queries, first, threads=8)
        assert np.array_equal(bok, o_bok)
        got = np.stack([bd["forward"]["node"], bd["forward"]["start"], bd["forward"]["end"],
                        bd["reverse"]["node"], bd["reverse"]["start"], bd["reverse"]["end"]], axis=1)
        assert np.array_equal(got[bok], o_bd[o_bok])
        # a found bidirectional state has equal forward / reverse range lengths and the right end nodes
        assert np.array_equal(got[bok][:, 2] - got[bok][:, 1], got[bok][:, 5] - got[bok][:, 4])
        assert np.array_equal(got[bok][:, 0], queries[bok][:, 9]) and np.array_equal(got[bok][:, 3], queries[bok][:, 0] ^ 1)


def test_large_query_batches_travel_in_chunks(monkeypatch):
    """Large host-pointer query calls (1.3 M rows, tens of MB each way) -- every entry point of the navigation / search group: the same
    answers from two workspaces of one handle, from the device-resident forms, and (a seeded sample) from the oracle.  (Round 5 moved such
    batches in chunks through pinned copy lanes behind GBWT_HIP_QUERY_PIPELINE; measured slower than the one pageable copy each way and
    removed in round 6: profiles/r06_download_probe.txt.)"""
    import torch
    s = S.Synth.chain(sites=2000, haplotypes=600, alleles=2, model=S.MOSAIC, founders=16, switch_rate=5e-3, seed=29)
    dev, oracle = open_synth(s), oracle_of(s)
    plain = dev.another_workspace()
    gen = np.random.default_rng(12)
    n = 1300003
    nodes = gen.integers(0, s.alphabet_size + 3, n, dtype=np.uint64)
    a, a_ok = dev.find(nodes)
    b, b_ok = plain.find(nodes)
    assert np.array_equal(a, b) and np.array_equal(a_ok, b_ok) and 0.3 < a_ok.mean() < 1.0
    for k in gen.choice(n, 300, replace=False):
        exp = oracle.find(int(nodes[k]))
        assert bool(a_ok[k]) == (exp is not None) and (exp is None or tuple(int(v) for v in a[k]) == exp)
    nxt = nodes + np.where(nodes % 6 < 2, 2, 4).astype(np.uint64)                  # anchor -> an allele, allele -> the next anchor (or nothing)
    e, e_ok = dev.extend(a, nxt)
    f, f_ok = plain.extend(a, nxt)
    assert np.array_equal(e, f) and np.array_equal(e_ok, f_ok) and e_ok.any()
    for k in gen.choice(np.flatnonzero(a_ok), 300, replace=False):
        exp = oracle.extend(tuple(int(v) for v in a[k]), int(nxt[k]))
        assert bool(e_ok[k]) == (exp is not None) and (exp is None or tuple(int(v) for v in e[k]) == exp)
    bd, bd_ok = dev.bd_find(nodes)
    bd1, bd1_ok = plain.bd_find(nodes)
    assert np.array_equal(bd, bd1) and np.array_equal(bd_ok, bd1_ok)
    x, x_ok = dev.extend_forward(bd, nxt)
    y, y_ok = plain.extend_forward(bd, nxt)
    assert np.array_equal(x, y) and np.array_equal(x_ok, y_ok)
    x, x_ok = dev.extend_backward(bd, nodes - np.uint64(2))
    y, y_ok = plain.extend_backward(bd, nodes - np.uint64(2))
    assert np.array_equal(x, y) and np.array_equal(x_ok, y_ok)
    pos = np.zeros(n, dtype=G.POS_DTYPE)
    pos["node"], pos["offset"] = nodes, gen.integers(0, 700, n)
    p, p_ok = dev.forward(pos)
    q, q_ok = plain.forward(pos)
    assert np.array_equal(p, q) and np.array_equal(p_ok, q_ok) and p_ok.any() and not p_ok.all()
    p, p_ok = dev.backward(pos)
    q, q_ok = plain.backward(pos)
    assert np.array_equal(p, q) and np.array_equal(p_ok, q_ok)
    ids = gen.integers(0, s.sequences + 5, n, dtype=np.uint64)
    p, p_ok = dev.start(ids)
    q, q_ok = plain.start(ids)
    assert np.array_equal(p, q) and np.array_equal(p_ok, q_ok)
    # whole queries: chunked, in one piece, device-resident
    queries = make_queries(s, random.Random(5), 3000, 7)
    queries = np.ascontiguousarray(np.tile(queries, (60, 1))[:170001])
    st, ok = dev.search(queries)
    st1, ok1 = plain.search(queries)
    d_q = torch.from_numpy(queries.view(np.int64)).cuda()
    st2, ok2 = plain.states_to_host(plain.search_device(d_q.data_ptr(), len(queries), 7))
    assert np.array_equal(st, st1) and np.array_equal(ok, ok1) and np.array_equal(st, st2) and np.array_equal(ok, ok2)
    o_st, o_ok = oracle.search_batch(queries[:3000], threads=8)
    assert np.array_equal(ok[:3000], o_ok) and np.array_equal(np.stack([st["node"], st["start"], st["end"]], axis=1)[:3000][o_ok], o_st[o_ok])
    bd, bok = dev.bd_search(queries, 3)
    bd1, bok1 = plain.bd_search(queries, 3)
    bd2, bok2 = plain.states_to_host(plain.bd_search_device(d_q.data_ptr(), len(queries), 7, 3), bidirectional=True)
    assert np.array_equal(bd, bd1) and np.array_equal(bok, bok1) and np.array_equal(bd, bd2) and np.array_equal(bok, bok2)
    empty = plain.search_device(0, 0, 7)
    assert empty.n == 0
    # queries of no nodes at all: find(q[0]) has nothing to find -- "not found" for every row, through every way in (src/bin/benchmark.rs builds none)
    z2, z2_ok = plain.states_to_host(plain.search_device(d_q.data_ptr(), 5, 0))
    assert not z2_ok.any() and not z2["node"].any() and not z2["end"].any()
    z, z_ok = plain.search(np.zeros((5, 0), dtype=np.uint64))          # (rows of no bytes: the host form needs no buffer for them)
    assert not z_ok.any() and np.array_equal(z, z2)
    with pytest.raises(G.GbwtHipError):
        plain.search_device(0, 5, 7)                      # rows but no pointer
    plain.close()


def test_open_flags_build_what_they_name(tmp_path):
    """gbwt_hip_open_*_flags (round 5): a handle opened for SEARCH answers the navigation / search group bit for bit like a handle opened
    for everything, holds a fraction of its memory (no walk descriptors, two-step blocks, samples) and refuses extraction and GFA lines with
    GBWT_HIP_BAD_ARGUMENT; a handle opened for EXTRACT extracts like the full one and refuses searches and lines; GFA implies EXTRACT."""
    s = S.Synth.chain(sites=3000, haplotypes=800, alleles=2, model=S.MOSAIC, founders=16, switch_rate=5e-3, seed=31)
    path = str(tmp_path / "flags.gbz")
    s.save(path, as_gbz=True)
    full = G.GBZ.load(path)
    search = G.GBZ.load(path, flags=G.OPEN_SEARCH)
    extract = G.GBZ.load(path, flags=G.OPEN_EXTRACT)
    lines = G.GBZ.load(path, flags=G.OPEN_GFA)
    recs = G.GBWT.from_records(s.data(), s.starts(), s.alphabet_offset, s.alphabet_size, s.sequences, s.size, True, flags=G.OPEN_SEARCH)
    mem = {k: h.memory_usage()["index_device_bytes"] for k, h in (("full", full), ("search", search), ("extract", extract), ("lines", lines))}
    assert mem["search"] < 0.5 * mem["full"] and mem["extract"] < mem["full"] and mem["extract"] <= mem["lines"] <= mem["full"], mem
    queries = make_queries(s, random.Random(9), 4000, 8)
    a, a_ok = full.search(queries)
    for h in (search, recs):
        b, b_ok = h.search(queries)
        assert np.array_equal(a, b) and np.array_equal(a_ok, b_ok)
        bd, bd_ok = h.bd_search(queries, 3)
        fbd, fbd_ok = full.bd_search(queries, 3)
        assert np.array_equal(bd, fbd) and np.array_equal(bd_ok, fbd_ok)
        pos, ok = h.start(np.arange(s.sequences))
        fpos, fok = full.start(np.arange(s.sequences))
        assert np.array_equal(pos, fpos) and np.array_equal(ok, fok)
        nxt, nok = h.forward(pos)
        fnxt, fnok = full.forward(fpos)
        assert np.array_equal(nxt, fnxt) and np.array_equal(nok, fnok)
        back, bok = h.backward(nxt)
        fback, fbok = full.backward(fnxt)
        assert np.array_equal(back, fback) and np.array_equal(bok, fbok)
        with pytest.raises(G.GbwtHipError) as e:
            h.sequences_csr(np.arange(4))
        assert e.value.status == _lib_status("BAD_ARGUMENT")
    ids = np.arange(s.sequences, dtype=np.uint64)
    f_off, f_nodes = full.sequences_csr(ids)
    for h in (extract, lines):
        off, nodes = h.sequences_csr(ids)
        assert np.array_equal(off, f_off) and np.array_equal(nodes, f_nodes)
        with pytest.raises(G.GbwtHipError) as e:
            h.find([4])
        assert e.value.status == _lib_status("BAD_ARGUMENT")
    # ... and a handle without SEARCH whose walks never need the generic decoder has given its raw descriptors back (64 bytes per record);
    # what would read them -- the pool-output walk modes -- is refused, not answered wrongly
    records = int(full.stats.records)
    assert mem["extract"] <= mem["full"] - 128 * records, (mem, records)
    extract.tune(walk_mode=1)
    with pytest.raises(G.GbwtHipError) as e:
        extract.sequences_csr(ids[:8])
    assert e.value.status == _lib_status("UNSUPPORTED")
    extract.tune(walk_mode=0)
    off, nodes = extract.sequences_csr(ids[:8])
    assert np.array_equal(nodes, f_nodes[:int(f_off[8])])
    assert extract.sequences_csr(np.zeros(0, dtype=np.uint64))[0].tolist() == [0]
    assert lines.path_lines([0, 5, 7], 1) == full.path_lines([0, 5, 7], 1)
    for h in (search, extract):
        with pytest.raises(G.GbwtHipError):
            h.path_lines([0], 1)
    with pytest.raises(G.GbwtHipError):
        G.GBZ.load(path, flags=0)
    with pytest.raises(G.GbwtHipError):
        G.GBZ.load(path, flags=8)


def _lib_status(name):
    from gbwt_rs_amd import _lib
    return _lib.STATUS_NAMES.index(name)


def test_config_c3_scale_states_against_the_oracle():
    """Config 3 at a third of its full length (400 000 sites x 5 008 haplotypes, 1.2 M nodes; the full 1.1 M-site run is tools/search_bench.py):
    a million queries go through the device, EVERY final state of a seeded sample of 30 000 of them is compared with the oracle
    (unidirectional and bidirectional), and all of them satisfy the invariants a found state must satisfy."""
    s = S.Synth.chain(sites=400000, haplotypes=5008, alleles=2, model=S.MOSAIC, founders=32, switch_rate=2e-3, seed=22)
    dev, oracle = open_synth(s), oracle_of(s)
    gen = np.random.default_rng(8)                                         # queries as make_queries cuts them, vectorised: a million of them
    some = np.stack([s.path(int(h)) for h in gen.choice(s.paths, 48, replace=False)]).astype(np.uint64)
    which, at = gen.integers(0, len(some), 1000000), gen.integers(0, some.shape[1] - 10, 1000000)
    queries = some[which[:, None], at[:, None] + np.arange(10)]
    flip = gen.random(len(queries)) < 0.5
    queries[flip] = (queries[flip] ^ np.uint64(1))[:, ::-1]
    bad = np.flatnonzero(gen.random(len(queries)) < 0.05)
    queries[bad, gen.integers(0, 10, len(bad))] ^= np.uint64(2)
    queries = np.ascontiguousarray(queries)
    st, ok = dev.search(queries)
    bd, bok = dev.bd_search(queries, 4)
    assert 0.5 < ok.mean() <= 1.0 and np.array_equal(ok, bok)            # found one way = found the other way
    assert np.array_equal(st["node"][ok], queries[ok][:, 9])
    assert np.array_equal((st["end"] - st["start"])[ok], (bd["forward"]["end"] - bd["forward"]["start"])[ok])
    pick = np.sort(np.random.default_rng(8).choice(len(queries), 30000, replace=False))
    o_st, o_ok = oracle.search_batch(queries[pick], threads=8)
    assert np.array_equal(ok[pick], o_ok)
    got = np.stack([st["node"], st["start"], st["end"]], axis=1)[pick]
    assert np.array_equal(got[o_ok], o_st[o_ok])
    o_bd, o_bok = oracle.bd_search_batch(queries[pick], 4, threads=8)
    assert np.array_equal(bok[pick], o_bok)
    gbd = np.stack([bd["forward"]["node"], bd["forward"]["start"], bd["forward"]["end"],
                    bd["reverse"]["node"], bd["reverse"]["start"], bd["reverse"]["end"]], axis=1)[pick]
    assert np.array_equal(gbd[o_bok], o_bd[o_bok])


def test_two_threads_two_workspaces_one_index(monkeypatch):
    """A handle is shared by host threads, each with a workspace of its own (include/gbwt_hip.h; the reference shares &GBZ across
    rayon workers, src/bin/gbunzip.rs:421-434): two threads extract different batches from one index at the same time, again and again --
    one of them through the pool-output kernel, which makes it the one that builds the full-width two-step blocks on first use while
    the other is already walking -- and a third searches.  Every result against the generator / the oracle."""
    import threading
    s = S.Synth.chain(sites=20000, haplotypes=900, alleles=2, model=S.MOSAIC, founders=16, switch_rate=5e-3, seed=31, extra=1, indel_every=5)
    dev, oracle = open_synth(s), oracle_of(s)
    views = [dev.another_workspace()]
    monkeypatch.setenv("GBWT_HIP_DIRECT", "0")                             # the knobs are read when a workspace is created: this one takes the pool-output kernel
    views.append(dev.another_workspace())
    monkeypatch.delenv("GBWT_HIP_DIRECT")
    errors = []
    truth = [s.path_checksum(h) for h in range(s.paths)]

    def extract(view, first, rounds, pool):
        try:
            for r in range(rounds):
                ids = np.arange(first + (r % 3), s.sequences, 2 if not pool else 6, dtype=np.uint64)
                offsets, nodes = view.sequences_csr(ids)
                for k in (0, len(ids) // 2, len(ids) - 1):
                    row = nodes[offsets[k]:offsets[k + 1]]
                    p = int(ids[k]) // 2
                    exp = s.path(p) if ids[k] % 2 == 0 else (s.path(p) ^ 1)[::-1]
                    if not np.array_equal(row, exp):
                        errors.append(("row", first, r, k))
                if first == 0 and r % 3 == 0:
                    out = view.extract_device(ids)
                    sums = view.path_sums(len(ids))
                    if any(int(sums[k]) != truth[int(ids[k]) // 2] for k in range(len(ids))) or int(out.total) != int(offsets[-1]):
                        errors.append(("sums", r))
        except Exception as e:                                              # noqa: BLE001
            errors.append(repr(e))

    def search():
        try:
            rng = random.Random(1)
            for _ in range(6):
                queries = make_queries(s, rng, 4000, 8)
                st, ok = dev.search(queries)
                o_st, o_ok = oracle.search_batch(queries, threads=2)
                if not (np.array_equal(ok, o_ok) and np.array_equal(np.stack([st["node"], st["start"], st["end"]], axis=1)[ok], o_st[o_ok])):
                    errors.append("search")
        except Exception as e:                                              # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=extract, args=(views[0], 0, 9, False)), threading.Thread(target=extract, args=(views[1], 1, 6, True)),
               threading.Thread(target=search)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    for v in views:
        v.close()


@pytest.mark.parametrize("env", [{"GBWT_HIP_DIRECT": "0"}, {"GBWT_HIP_BOTH_ENDS": "0"}, {"GBWT_HIP_SEQ_LEN": "0"}, {},
                                 {"GBWT_HIP_SAMPLE_INTERVAL": "0"},                                   # no samples: rows filled from both ends
                                 {"GBWT_HIP_SAMPLE_INTERVAL": "0", "GBWT_HIP_BOTH_ENDS": "0"},      #             ... from one end
                                 {"GBWT_HIP_ORIENTATION_CHECK": "1", "GBWT_HIP_SEGMENTS": "0"}])    # samples present, both ends used
def test_extraction_output_paths(monkeypatch, env):
    """The extraction has three output paths: rows filled from both ends (default, bidirectional indexes whose sequence
    pairs check out), rows filled from one end, and the pool of chained blocks + compaction (no sequence lengths).
    Odd and even lengths, empty sequences, lengths below one ring chunk, duplicates, reverse sequences."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = random.Random(31)
    paths = [[2 * rng.randint(1, 40) + rng.randint(0, 1) for _ in range(ln)] for ln in (0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 64, 65, 127, 700, 1001)]
    s = S.Synth.from_paths(paths, bidirectional=True)
    dev = open_synth(s)
    ids = list(range(s.sequences)) + [5, 5, 28]
    offsets, nodes = dev.sequences_csr(ids)
    for k, i in enumerate(ids):
        exp = paths[i // 2] if i % 2 == 0 else kat.reverse_path(paths[i // 2])
        assert list(nodes[offsets[k]:offsets[k + 1]]) == exp, (i, env)
    chain = S.Synth.chain(sites=900, haplotypes=200, alleles=2, model=S.MOSAIC, founders=8, switch_rate=0.02, seed=23)
    cdev, coracle = open_synth(chain), oracle_of(chain)
    cids = np.arange(0, chain.sequences, dtype=np.uint64)
    o_off, o_nodes = coracle.extract(cids, threads=4)
    c_off, c_nodes = cdev.sequences_csr(cids)
    assert np.array_equal(c_off, o_off) and np.array_equal(c_nodes, o_nodes)


SEGMENT_ENVS = [{},                                                             # defaults: 4096-node segments, 128-byte row pieces
                {"GBWT_HIP_SAMPLE_INTERVAL": "8"},                              # the shortest segments there are
                {"GBWT_HIP_SAMPLE_INTERVAL": "16", "GBWT_HIP_ROW_PIECE": "16"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "37", "GBWT_HIP_RING_SLOTS": "128"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "64", "GBWT_HIP_UNIFORM_LOOP": "0"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "100", "GBWT_HIP_ROW_PIECE": "0"},     # every lane writes its own row
                {"GBWT_HIP_SAMPLE_INTERVAL": "250", "GBWT_HIP_XCD_MAP": "0"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "128", "GBWT_HIP_WIDE_ADDRESSES": "1"},   # both loops with 64-bit addresses
                {"GBWT_HIP_SAMPLE_INTERVAL": "1000", "GBWT_HIP_PATHS_PER_WAVE": "13"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "32", "GBWT_HIP_WALK_TABLES": "0"},   # outdegree > 2: plain table steps, one at a time
                {"GBWT_HIP_SAMPLE_INTERVAL": "48", "GBWT_HIP_DEEP_TABLES": "0"},   # ... walk-table steps, one per load instead of seven
                {"GBWT_HIP_SAMPLE_INTERVAL": "96", "GBWT_HIP_COMPACT_TABLES": "0"},   # ... seven per load everywhere (no twelve-step compact entries)
                {"GBWT_HIP_SAMPLE_INTERVAL": "24", "GBWT_HIP_SERIAL_SAMPLES": "1", "GBWT_HIP_TWO_PASS_OPEN": "1"},  # every sequence walked at open: lengths, then samples
                {"GBWT_HIP_SAMPLE_INTERVAL": "40", "GBWT_HIP_SERIAL_SAMPLES": "1"},   # ... both in one walk (samples every 40 nodes of each sequence)
                {"GBWT_HIP_SAMPLE_INTERVAL": "16", "GBWT_HIP_CHECKPOINT_CAP": "5"},   # checkpoint sampling with hops of at most 5 + 3 nodes: many rounds of orphans
                {"GBWT_HIP_SAMPLE_INTERVAL": "2048", "GBWT_HIP_CHECKPOINT_CAP": "100000"},   # ... with hardly any checkpoint: whole sequences in one hop
                {"GBWT_HIP_SAMPLE_INTERVAL": "90", "GBWT_HIP_CHAINS": "0"},        # no chained steps: fused pairs only
                {"GBWT_HIP_SAMPLE_INTERVAL": "19", "GBWT_HIP_CHAINS": "2", "GBWT_HIP_RING_SLOTS": "32", "GBWT_HIP_ROW_PIECE": "16"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "200", "GBWT_HIP_CATCH_UP": "0"},     # mixed waves go to the gather loop at once (no single steps of the lanes behind)
                {"GBWT_HIP_SAMPLE_INTERVAL": "200", "GBWT_HIP_CATCH_UP": "2"},     # the single steps on the two-step descriptors + packed half-blocks (what a lean handle has, round 5)
                {"GBWT_HIP_SAMPLE_INTERVAL": "16", "GBWT_HIP_CATCH_UP": "2", "GBWT_HIP_CHAINS": "0"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "64", "GBWT_HIP_CATCH_UP": "2", "GBWT_HIP_GATHER_LIMIT": "64"},   # ... with records too long for the packed counts in the way
                {"GBWT_HIP_SAMPLE_INTERVAL": "100", "GBWT_HIP_ROW_PIECE": "0"},    # the lane-per-row writer (no cooperative row pieces)
                {"GBWT_HIP_SAMPLE_INTERVAL": "64", "GBWT_HIP_ROW_PIECE": "16"},    # 64-byte row pieces
                {"GBWT_HIP_SAMPLE_INTERVAL": "8", "GBWT_HIP_SAMPLE_STRIDE": "2"},   # strided walkers (round 4): a walker per 2 / 3 / 5 samples of a row
                {"GBWT_HIP_SAMPLE_INTERVAL": "16", "GBWT_HIP_SAMPLE_STRIDE": "3"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "24", "GBWT_HIP_SAMPLE_STRIDE": "5"},
                {"GBWT_HIP_SAMPLE_INTERVAL": "64", "GBWT_HIP_SAMPLE_STRIDE": "1000"},   # ... more than any row has: one walker per row
                {"GBWT_HIP_SEGMENTS": "0"}]                                      # samples present but unused: one walker per end


def _skipping_haplotypes(sites, haplotypes, seed):
    """Haplotypes over a bubble chain that skip whole stretches of sites: the rows of a wave have different lengths, so
    walkers of the same segment number sit on different records (non-uniform waves, with uniform stretches in between)."""
    rng = random.Random(seed)
    paths = []
    for h in range(haplotypes):
        p, site = [], 0
        while site < sites:
            p.append(2 * (3 * site + 1))
            p.append(2 * (3 * site + 2 + (rng.random() < 0.3)))
            site += 1 if rng.random() < 0.97 else rng.randint(2, 9)
        paths.append(p if h % 11 else p[:rng.randint(0, len(p))])
    return paths


@pytest.mark.parametrize("env", SEGMENT_ENVS, ids=lambda e: ",".join(f"{k[9:]}={v}" for k, v in e.items()) or "defaults")
def test_segmented_extraction(monkeypatch, env):
    """Segmented extraction (sequence samples at open, one walker per sample interval of every row, cooperative row
    writes by the helper wave, wave-uniform loop with scalar descriptor fetch) against the oracle for every knob: rows of
    every length around the piece and ring sizes, empty rows, duplicates, reverse sequences, lock-step bubble chains
    (uniform waves), haplotypes that drift apart (mixed waves), and outdegree > 2 (slow records leave the hot loops)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = random.Random(41)
    lengths = [0, 1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 700, 1001, 4097]
    paths = [[2 * rng.randint(1, 40) + rng.randint(0, 1) for _ in range(ln)] for ln in lengths]
    s = S.Synth.from_paths(paths, bidirectional=True)
    dev = open_synth(s)
    ids = list(range(s.sequences)) + [5, 5, 28, 43]
    offsets, nodes = dev.sequences_csr(ids)
    for k, i in enumerate(ids):
        exp = paths[i // 2] if i % 2 == 0 else kat.reverse_path(paths[i // 2])
        assert list(nodes[offsets[k]:offsets[k + 1]]) == exp, (i, env)
    cases = [S.Synth.chain(sites=900, haplotypes=200, alleles=2, model=S.MOSAIC, founders=8, switch_rate=0.02, seed=23),
             S.Synth.chain(sites=300, haplotypes=130, alleles=5, model=S.IID, zipf=0.5, seed=24),
             S.Synth.from_paths(_skipping_haplotypes(400, 150, 25), bidirectional=True)]
    for c in cases:
        cdev, coracle = open_synth(c), oracle_of(c)
        for cids in (np.arange(0, c.sequences, dtype=np.uint64), np.arange(0, c.sequences, 2, dtype=np.uint64)[::-1].copy()):
            o_off, o_nodes = coracle.extract(cids, threads=4)
            c_off, c_nodes = cdev.sequences_csr(cids)
            assert np.array_equal(c_off, o_off) and np.array_equal(c_nodes, o_nodes), env


@pytest.mark.parametrize("env", [{}, {"GBWT_HIP_SAMPLE_INTERVAL": "8"}, {"GBWT_HIP_SAMPLE_INTERVAL": "50", "GBWT_HIP_XCD_MAP": "0"},
                                 {"GBWT_HIP_SAMPLE_INTERVAL": "8", "GBWT_HIP_SAMPLE_STRIDE": "3"}, {"GBWT_HIP_SAMPLE_INTERVAL": "16", "GBWT_HIP_SAMPLE_STRIDE": "2", "GBWT_HIP_ROW_PIECE": "16"}],
                         ids=lambda e: ",".join(f"{k[9:]}={v}" for k, v in e.items()) or "defaults")
def test_walker_order_with_ragged_rows(monkeypatch, env):
    """A few long haplotypes and thousands of short walks (a fragmented assembly): the walkers of a segmented extraction
    are numbered segment by segment over the rows that HAVE the segment (rows sorted by segment count, level prefix
    sums), in every batch order, with duplicates and empty rows; compared with the input paths."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = random.Random(77)

    def walk(first, count):
        p = []
        for site in range(first, first + count):
            p += [2 * (3 * site + 1), 2 * (3 * site + 2 + (rng.random() < 0.3))]
        return p

    sites = 5000
    paths = [walk(0, sites), walk(100, sites - 100), walk(0, sites // 3)] + [walk(rng.randrange(0, sites - 60), rng.randint(0, 60)) for _ in range(3000)]
    s = S.Synth.from_paths(paths, bidirectional=True)
    dev = open_synth(s)
    batches = [list(range(0, s.sequences, 2)), list(range(s.sequences - 1, -1, -1)), [rng.randrange(s.sequences) for _ in range(777)] + [0, 0, 1]]
    for ids in batches:
        offsets, nodes = dev.sequences_csr(ids)
        assert len(offsets) == len(ids) + 1
        for k, i in enumerate(ids):
            exp = paths[i // 2] if i % 2 == 0 else kat.reverse_path(paths[i // 2])
            assert np.array_equal(nodes[offsets[k]:offsets[k + 1]], np.array(exp, dtype=np.uint32)), (i, env)


PART_ENVS = [{}, {"GBWT_HIP_SAMPLE_INTERVAL": "8"}, {"GBWT_HIP_SAMPLE_INTERVAL": "16", "GBWT_HIP_SAMPLE_STRIDE": "3"},
             {"GBWT_HIP_SAMPLE_INTERVAL": "24", "GBWT_HIP_ROW_PIECE": "16"},
             {"GBWT_HIP_SAMPLE_INTERVAL": "0"}, {"GBWT_HIP_SEGMENTS": "0"}, {"GBWT_HIP_SAMPLE_INTERVAL": "32", "GBWT_HIP_DEFER_TOTAL": "0"},
             # the pool-output fallbacks cannot cut rows: the whole row is the LAST part, every earlier part is empty (never the row from every part)
             {"GBWT_HIP_DIRECT": "0"}, {"GBWT_HIP_WALK_MODE": "1"}, {"GBWT_HIP_SEQ_LEN": "0"}]


@pytest.mark.parametrize("env", PART_ENVS, ids=lambda e: ",".join(f"{k[9:]}={v}" for k, v in e.items()) or "defaults")
def test_parts_of_rows(monkeypatch, env):
    """gbwt_hip_extract_part_device: every row cut at sequence samples into `parts` stretches, one stretch of EVERY row per call (what one rank
    of a multi-GPU extraction walks).  The stretches of a row, in order, are the row: against the oracle, for 1, 2, 3, 8 and more parts than
    any row has samples, with rows of every length (empty ones, rows of one sample), duplicates, ids without a sequence, reverse sequences,
    lock-step and ragged batches, every walker order, and indexes without samples (the last part is the whole row there)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = random.Random(43)
    lengths = [0, 1, 2, 7, 9, 16, 31, 33, 64, 65, 129, 255, 700, 1001, 4097]
    paths = [[2 * rng.randint(1, 40) + rng.randint(0, 1) for _ in range(ln)] for ln in lengths]
    ragged = S.Synth.from_paths(paths, bidirectional=True)
    cases = [(ragged, list(range(ragged.sequences)) + [5, 5, 28, ragged.sequences + 3, 1]),
             (S.Synth.chain(sites=900, haplotypes=200, alleles=2, model=S.MOSAIC, founders=8, switch_rate=0.02, seed=23), None),
             (S.Synth.chain(sites=300, haplotypes=130, alleles=5, model=S.IID, zipf=0.5, seed=24), None),
             (S.Synth.from_paths(_skipping_haplotypes(400, 150, 25), bidirectional=True), None)]
    for c, ids in cases:
        dev = open_synth(c)
        batches = [np.array(ids, dtype=np.uint64)] if ids is not None else [np.arange(0, c.sequences, 2, dtype=np.uint64), np.arange(c.sequences - 1, -1, -1, dtype=np.uint64)]
        for batch in batches:
            w_off, w_nodes = dev.sequences_csr(batch)
            if ids is None:
                o_off, o_nodes = oracle_of(c).extract(batch, threads=4)
                assert np.array_equal(w_off, o_off) and np.array_equal(w_nodes, o_nodes), env
            for parts in (1, 2, 3, 8, 5000):
                pieces = [dev.part_csr(batch, r, parts) for r in range(parts if parts < 100 else 0)] or \
                         [dev.part_csr(batch, r, parts) for r in (0, 1, 2499, 2500, 4998, 4999)]
                if parts >= 100:
                    # too many to join: every stretch is a piece of its row, where the stretches before it say (here: only that nothing is lost at the ends)
                    assert all(int(off[-1]) <= int(w_off[-1]) for off, _ in pieces), (env, parts)
                    total = sum(int(dev.extract_part_device(batch, r, parts).total) for r in range(parts))
                    assert total == int(w_off[-1]), (env, parts)
                    continue
                at = np.zeros(len(batch), dtype=np.uint64)
                for r, (off, nodes) in enumerate(pieces):
                    assert len(off) == len(batch) + 1, (env, parts)
                    for k in range(len(batch)):
                        ln = int(off[k + 1] - off[k])
                        lo = int(w_off[k] + at[k])
                        assert lo + ln <= int(w_off[k + 1]) and np.array_equal(nodes[int(off[k]):int(off[k + 1])], w_nodes[lo:lo + ln]), (env, parts, r, k)
                        at[k] += ln
                assert np.array_equal(at, np.diff(w_off)), (env, parts)
    with pytest.raises(Exception):
        dev.extract_part_device(batch, 3, 3)


@pytest.mark.parametrize("defer", ["1", "0"])
def test_rows_sized_after_the_launch(monkeypatch, defer):
    """A request that knows its walkers without the device launches the walk into the rows its workspace has and looks at the total
    afterwards (GBWT_HIP_DEFER_TOTAL, round 4): requests that grow (the rows are too small: nobody walks, the host makes them and launches
    again), shrink, come as parts and as whole rows, in one workspace; and batches of more rows than the one-launch row offsets take."""
    monkeypatch.setenv("GBWT_HIP_DEFER_TOTAL", defer)
    monkeypatch.setenv("GBWT_HIP_SAMPLE_INTERVAL", "64")
    c = S.Synth.chain(sites=700, haplotypes=300, alleles=3, model=S.MOSAIC, founders=6, switch_rate=0.01, seed=77)
    dev, oracle = open_synth(c), oracle_of(c)
    forward = np.arange(0, c.sequences, 2, dtype=np.uint64)
    for ids in (forward[:3], forward[:40], forward, forward[:7], np.arange(c.sequences, dtype=np.uint64), forward[::-1].copy()):
        o_off, o_nodes = oracle.extract(ids, threads=4)
        for parts in (1, 4, 1, 2):
            at = np.zeros(len(ids), dtype=np.uint64)
            for r in range(parts):
                off, nodes = dev.part_csr(ids, r, parts)
                for k in range(len(ids)):
                    ln, lo = int(off[k + 1] - off[k]), int(o_off[k] + at[k])
                    assert np.array_equal(nodes[int(off[k]):int(off[k + 1])], o_nodes[lo:lo + ln]), (defer, parts, r, k)
                    at[k] += ln
            assert np.array_equal(at, np.diff(o_off)), (defer, parts)
    many = np.tile(forward, 900)[: (1 << 13) + 77]             # more rows than k_row_offsets takes: lengths + scan the long way
    off, nodes = dev.sequences_csr(many)
    o_off, o_nodes = oracle.extract(forward, threads=4)
    ln = np.diff(o_off)
    assert np.array_equal(np.diff(off), np.tile(ln, 900)[: len(many)])
    for k in (0, 299, 300, 8191, 8192, len(many) - 1):
        h = k % len(forward)
        assert np.array_equal(nodes[int(off[k]):int(off[k + 1])], o_nodes[int(o_off[h]):int(o_off[h + 1])])


def _layered_paths(layers, haplotypes, seed):
    """Paths over a layered graph with layers 1 .. 6 nodes wide, some layers skipped by some haplotypes: outdegrees from 1
    to 12, table records followed directly by table records, by unary records and by outdegree-2 records."""
    rng = random.Random(seed)
    widths = [rng.choice((1, 1, 2, 2, 3, 5, 6)) for _ in range(layers)]
    first = [1]
    for w in widths:
        first.append(first[-1] + w)
    paths = []
    for h in range(haplotypes):
        p = []
        for layer, w in enumerate(widths):
            if rng.random() < 0.05:
                continue
            node = first[layer] + min(w - 1, int(rng.random() ** 2 * w))
            p.append(2 * node + (rng.random() < 0.02))
        paths.append(p if h % 13 else p[:rng.randint(0, len(p))])
    return paths


@pytest.mark.parametrize("env", [{}, {"GBWT_HIP_SAMPLE_INTERVAL": "8"}, {"GBWT_HIP_SAMPLE_INTERVAL": "9", "GBWT_HIP_RING_SLOTS": "32", "GBWT_HIP_ROW_PIECE": "16"},
                                 {"GBWT_HIP_WALK_TABLES": "0"}, {"GBWT_HIP_SEGMENTS": "0"}, {"GBWT_HIP_DIRECT": "0"},
                                 {"GBWT_HIP_SAMPLE_INTERVAL": "11", "GBWT_HIP_CATCH_UP": "0"}, {"GBWT_HIP_SAMPLE_INTERVAL": "11", "GBWT_HIP_CATCH_UP": "2"},
                                 {"GBWT_HIP_DEEP_TABLES": "0"}, {"GBWT_HIP_SAMPLE_INTERVAL": "13", "GBWT_HIP_DEEP_TABLES": "0"},   # one table step per load
                                 {"GBWT_HIP_SAMPLE_INTERVAL": "10", "GBWT_HIP_RING_SLOTS": "128"}, {"GBWT_HIP_SAMPLE_INTERVAL": "21", "GBWT_HIP_ROW_PIECE": "0"},
                                 {"GBWT_HIP_SAMPLE_INTERVAL": "0"}],   # whole sequences from both ends: seven steps at a time past the middle of the row
                         ids=lambda e: ",".join(f"{k[9:]}={v}" for k, v in e.items()) or "defaults")
def test_walk_tables(monkeypatch, env):
    """Walks over records with outdegree > 2 (walk tables: the plain LF entry with the step through a unary successor and
    the landing record's table / block base folded in) in a graph where table records are followed by every kind of
    record, with segment boundaries of every phase (a boundary can fall between the two nodes of a fused entry), against
    the oracle; forward() on the same index still answers with the plain tables."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for seed, (layers, haplotypes) in enumerate([(60, 90), (300, 200), (37, 1500)]):
        paths = _layered_paths(layers, haplotypes, 50 + seed)
        s = S.Synth.from_paths(paths, bidirectional=True)
        dev, oracle = open_synth(s), oracle_of(s)
        assert dev.stats.max_outdegree > 2
        ids = np.arange(0, s.sequences, dtype=np.uint64)
        o_off, o_nodes = oracle.extract(ids, threads=4)
        offsets, nodes = dev.sequences_csr(ids)
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes), (env, seed)
        some = np.array([rng_id for rng_id in random.Random(seed).sample(range(s.sequences), min(29, s.sequences))], dtype=np.uint64)
        o_off, o_nodes = oracle.extract(some)
        offsets, nodes = dev.sequences_csr(some)
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes), (env, seed)
        if seed == 0:
            check_all_positions(dev, oracle)


def test_two_step_walk_with_64_bit_addresses(monkeypatch):
    """The two-step loop has two addressing variants (SGPR base + 32-bit offsets below 4 GiB, 64-bit addresses above);
    GBWT_HIP_WIDE_ADDRESSES forces the second one, which no test index is large enough to need."""
    s = S.Synth.chain(sites=700, haplotypes=300, alleles=2, model=S.MOSAIC, founders=8, switch_rate=0.02, seed=21)
    dev, oracle = open_synth(s), oracle_of(s)
    ids = np.arange(0, s.sequences, dtype=np.uint64)
    o_off, o_nodes = oracle.extract(ids, threads=4)
    monkeypatch.setenv("GBWT_HIP_WIDE_ADDRESSES", "1")
    offsets, nodes = dev.sequences_csr(ids)
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    monkeypatch.delenv("GBWT_HIP_WIDE_ADDRESSES")
    offsets, nodes = dev.sequences_csr(ids)
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)


def test_generic_records_without_lf_tables(monkeypatch):
    """Outdegree > 2 with the LF tables switched off (GBWT_HIP_TABLE_BYTES=0, read at open): the serial Record::lf decode
    behind the same walk frames, in every walk mode, and forward() through the generic scan."""
    monkeypatch.setenv("GBWT_HIP_TABLE_BYTES", "0")
    s = S.Synth.chain(sites=40, haplotypes=300, alleles=7, model=S.IID, zipf=0.5, seed=9)
    dev, oracle = open_synth(s), oracle_of(s)
    monkeypatch.delenv("GBWT_HIP_TABLE_BYTES")
    ids = np.arange(0, s.sequences, dtype=np.uint64)
    o_off, o_nodes = oracle.extract(ids, threads=4)
    for mode in (0, 3, 2, 1):
        dev.tune(mode, 0, 16)
        offsets, nodes = dev.sequences_csr(ids)
        assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes), mode
    dev.tune(0, 0, 16)
    check_all_positions(dev, oracle)


def test_high_degree_search_and_extract():
    """BASELINE config 5 shape (outdegree >= 255, two-varint runs): extraction and search stay bit-exact."""
    s = S.Synth.chain(sites=30, haplotypes=2500, alleles=300, model=S.IID, zipf=0.3, seed=5)
    dev, oracle = open_synth(s), oracle_of(s)
    assert dev.stats.max_outdegree >= 255
    ids = np.arange(0, s.sequences, dtype=np.uint64)
    offsets, nodes = dev.sequences_csr(ids)
    o_off, o_nodes = oracle.extract(ids, threads=8)
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    queries = make_queries(s, random.Random(8), 3000, 6)
    st, ok = dev.search(queries)
    o_st, o_ok = oracle.search_batch(queries, threads=8)
    assert np.array_equal(ok, o_ok)
    assert np.array_equal(np.stack([st["node"], st["start"], st["end"]], axis=1)[ok], o_st[o_ok])
    bd, bok = dev.bd_search(queries, 2)
    o_bd, o_bok = oracle.bd_search_batch(queries, 2, threads=8)
    assert np.array_equal(bok, o_bok)


def test_cpp_mirror_of_the_reference_tests(tmp_path):
    """tests/cpp/test_reference_api.cpp: the reference's own tests restated against include/gbwt_hip.hpp."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "test_reference_api"
    csrc = os.path.join(root, "gbwt_rs_amd", "csrc")
    subprocess.run([shutil.which("g++") or "g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(root, "include"), "-o", str(exe),
                    os.path.join(root, "tests", "cpp", "test_reference_api.cpp"), "-L", csrc, "-lgbwt_hip", "-Wl,-rpath," + csrc], check=True)
    out = subprocess.run([str(exe), GOLDEN], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout


def test_corrupt_files_never_take_the_device_down(tmp_path):
    """Every 64-bit element of the small fixtures overwritten with hostile values: whatever the host parser accepts is
    opened on the GPU and queried (extract everything, find + forward over all nodes).  Records full of garbage may give
    garbage, or GBWT_HIP_INVALID_DATA when a walk never ends -- never a fault, a hang or an abort (the reference panics
    on malformed records, src/bwt.rs:374-377; the C side returns a status).  Runs in a child process with a deadline."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gbwt_rs_amd as G
from test_capi_cpu import mutated_files
import pathlib
opened = rejected = 0
for name, w, value, path in mutated_files(pathlib.Path(TMP), names=("example.gbwt", "with-empty.gbwt", "translation.gbz"), values=(0xFFFFFFFFFFFFFFFF, 0, 65, 1 << 40, 0x0101010101010101)):
    try:
        dev = G.GBZ.load(path)
    except G.GbwtHipError as e:
        assert e.status in (G._lib.INVALID_DATA, G._lib.DEVICE_ERROR, G._lib.UNSUPPORTED), (name, w, hex(value), str(e))
        rejected += 1
        continue
    opened += 1
    n_seq = min(dev.sequences(), 1 << 12)
    off, nodes = dev.sequences_csr(np.arange(n_seq + 2, dtype=np.uint64))
    assert int(off[-1]) == len(nodes)
    hi = min(dev.alphabet_size() + 2, 1 << 12)
    st, ok = dev.find(np.arange(hi, dtype=np.uint64))
    pos = np.zeros(hi, dtype=G.POS_DTYPE); pos["node"] = np.arange(hi); pos["offset"] = 1
    dev.forward(pos)
    if dev.is_bidirectional():
        dev.backward(pos)
        bd, ok = dev.bd_find(np.arange(hi, dtype=np.uint64))
        dev.follow(bd[ok][:64])
    if dev.stats.is_gbz and dev.has_metadata() and dev.stats.paths:
        try:
            dev.path_lines(np.arange(min(dev.stats.paths, 8), dtype=np.uint64), 1)
        except G.GbwtHipError as e:
            assert e.status in (G._lib.BAD_ARGUMENT, G._lib.INVALID_DATA), str(e)
    dev.close()
print("opened", opened, "rejected", rejected)
assert opened > 300 and rejected > 300
print("MUTATIONS_OK")
'''
    out = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}; TMP = {str(tmp_path)!r}\n" + code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "MUTATIONS_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_corrupt_large_files_through_the_threaded_open(tmp_path):
    """The same for a GBZ of several MiB, which takes the threaded paths of gbwt_hip_open_file: the record bytes start for the device out
    of the mapped file while the loader still decodes (a decode that throws must not unmap the file under that copy: round 3 did), the
    deferred decodes run on threads, node labels and the host copy of the record bytes in the background behind finish().  Corrupt
    starts, labels, lengths: GBWT_HIP_INVALID_DATA (or an index that opens and answers), never a fault -- and never a stall: records that send
    the walks of the open in circles share one step budget (walk_loops.hpp: quiet_walk; round 5: two such mutations took three minutes each),
    so no open takes more than ten seconds.  Child process with a deadline."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gbwt_rs_amd as G
from test_capi_cpu import mutated_large_file
import pathlib
import time
opened = rejected = 0
reasons = set()
slowest = 0.0
for w, value, path in mutated_large_file(pathlib.Path(TMP), per_region=60):
    t0 = time.perf_counter()
    try:
        dev = G.GBZ.load(path)
        slowest = max(slowest, time.perf_counter() - t0)
    except G.GbwtHipError as e:
        slowest = max(slowest, time.perf_counter() - t0)
        assert e.status in (G._lib.INVALID_DATA, G._lib.UNSUPPORTED), (w, hex(value), str(e))
        rejected += 1
        reasons.add(str(e)[:60])
        continue
    opened += 1
    off, nodes = dev.sequences_csr(np.arange(0, min(dev.sequences(), 64), dtype=np.uint64))
    assert int(off[-1]) == len(nodes)
    if dev.stats.is_gbz and dev.has_metadata() and dev.stats.paths:
        try:
            dev.path_lines(np.arange(min(dev.stats.paths, 4), dtype=np.uint64), 1)
        except G.GbwtHipError as e:
            assert e.status in (G._lib.BAD_ARGUMENT, G._lib.INVALID_DATA), str(e)
    dev.close()
print("opened", opened, "rejected", rejected, sorted(reasons))
assert opened > 30 and rejected > 60 and any("starts" in r or "bitvector" in r or "Elias" in r or "sparse" in r.lower() for r in reasons), reasons
assert slowest < 30.0, f"an open of a corrupt file took {slowest:.1f} s"
print("MUTATIONS_OK")
'''
    out = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}; TMP = {str(tmp_path)!r}\n" + code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "MUTATIONS_OK" in out.stdout, (out.stdout[-800:], out.stderr[-3000:])
