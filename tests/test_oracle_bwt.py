"""Oracle BWT/Record vs the paper-example known answers; mirrors src/bwt/tests.rs:105-341."""
import pytest

import kat
import oracle_lib as O


def create_bwt(edges, runs):
    return O.OracleBWT(edges, runs)


def check_records(bwt, edges):  # src/bwt/tests.rs:105-136
    assert len(bwt) == len(edges)
    for i in range(len(bwt)):
        rec = bwt.record(i)
        assert (rec is None) == (len(edges[i]) == 0)
        if rec is not None:
            assert rec.outdegree == len(edges[i])
            assert rec.edges() == edges[i]
        comp = bwt.compressed_record(i)
        assert (comp is None) == (len(edges[i]) == 0)
        if comp is not None:
            edge_bytes, bwt_bytes = comp
            vals = O.bytecode_decode(edge_bytes)
            assert vals[0] == len(edges[i]) and len(vals) == 1 + 2 * len(edges[i])
            assert bwt_bytes == rec.bwt_bytes()


def check_lf(bwt, edges, runs):  # src/bwt/tests.rs:159-184
    for i in range(len(bwt)):
        rec = bwt.record(i)
        if rec is None:
            continue
        offset = 0
        cur = [list(e) for e in edges[i]]
        dec = rec.decompress()
        assert len(dec) == rec.len()
        for value, ln in runs[i]:
            for _ in range(ln):
                edge = tuple(cur[value])
                expected = None if edge[0] == kat.ENDMARKER else edge
                assert rec.lf(offset) == expected
                assert dec[offset] == edge
                assert rec.offset_to(edge) == (None if edge[0] == kat.ENDMARKER else offset)
                offset += 1
                cur[value][1] += 1
        assert rec.len() == offset
        assert rec.lf(offset) is None


def check_follow(bwt, invalid_node):  # src/bwt/tests.rs:189-237
    for i in range(len(bwt)):
        rec = bwt.record(i)
        if rec is None:
            continue
        ln = rec.len()
        succs = [e[0] for e in rec.edges()]
        for start in range(ln + 1):
            for limit in range(start, ln + 1):
                assert rec.follow(start, limit, kat.ENDMARKER) is None
                assert rec.bd_follow(start, limit, kat.ENDMARKER) is None
                for successor in succs:
                    if successor == kat.ENDMARKER:
                        continue
                    result = rec.follow(start, limit, successor)
                    if result is not None:
                        found = [result[0], result[0]]
                        for j in range(start, limit):
                            pos = rec.lf(j)
                            if pos is not None and pos[0] == successor and pos[1] == found[1]:
                                found[1] += 1
                        assert tuple(found) == result
                        bd = rec.bd_follow(start, limit, successor)
                        assert bd is not None and bd[0] == result
                    else:
                        for j in range(start, limit):
                            pos = rec.lf(j)
                            if pos is not None:
                                assert pos[0] != successor
                        assert rec.bd_follow(start, limit, successor) is None
                assert rec.follow(start, limit, invalid_node) is None
                assert rec.bd_follow(start, limit, invalid_node) is None


def negative_offset_to(bwt, invalid_node):  # src/bwt/tests.rs:240-257
    for i in range(len(bwt)):
        rec = bwt.record(i)
        if rec is None:
            continue
        assert rec.offset_to((kat.ENDMARKER, 0)) is None
        assert rec.offset_to((invalid_node, 0)) is None
        for successor, offset in rec.edges():
            if successor == kat.ENDMARKER:
                continue
            if offset > 0:
                assert rec.offset_to((successor, offset - 1)) is None
            r = rec.follow(0, rec.len(), successor)
            assert rec.offset_to((successor, offset + (r[1] - r[0]))) is None


def check_predecessor_at(bwt):  # src/bwt/tests.rs:261-283
    end = bwt.record(kat.ENDMARKER)
    starting = {end.lf(i) for i in range(end.len())}
    for i in range(1, len(bwt)):
        rec = bwt.record(i)
        if rec is None:
            continue
        reverse_id = ((i + 1) ^ 1) - 1
        rrec = bwt.record(reverse_id)
        for j in range(rec.len()):
            if (i + 1, j) in starting:
                assert rrec.predecessor_at(j) is None
            else:
                assert rrec.predecessor_at(j) is not None
        assert rrec.predecessor_at(rec.len()) is None


def test_empty_bwt():
    bwt = create_bwt([], [])
    check_records(bwt, [])
    assert len(bwt) == 0 and bwt.record(0) is None


def test_non_empty_bwt():
    bwt = create_bwt(kat.PAPER_EDGES, kat.PAPER_RUNS)
    check_records(bwt, kat.PAPER_EDGES)
    check_lf(bwt, kat.PAPER_EDGES, kat.PAPER_RUNS)
    check_follow(bwt, kat.PAPER_INVALID)
    negative_offset_to(bwt, kat.PAPER_INVALID)
    # doc-test src/bwt.rs:11-39: node 2: outdegree 2, lf(1) = (5, 0), follow(0..2, 5) = 0..1, total length 17
    rec = bwt.record(2)
    assert rec.outdegree == 2 and rec.lf(1) == (5, 0) and rec.follow(0, 2, 5) == (0, 1)
    assert sum(bwt.record(i).len() for i in range(len(bwt))) == 17


def test_empty_records():
    edges = [list(e) for e in kat.PAPER_EDGES]
    runs = [list(r) for r in kat.PAPER_RUNS]
    for k in (2, 6):
        edges[k], runs[k] = [], []
    bwt = create_bwt(edges, runs)
    check_records(bwt, edges)
    check_lf(bwt, edges, runs)
    check_follow(bwt, kat.PAPER_INVALID)
    negative_offset_to(bwt, kat.PAPER_INVALID)
    assert bwt.record_bytes(2) == b"\x00"


def test_bidirectional_bwt():
    bwt = create_bwt(kat.BD_EDGES, kat.BD_RUNS)
    check_records(bwt, kat.BD_EDGES)
    check_lf(bwt, kat.BD_EDGES, kat.BD_RUNS)
    check_follow(bwt, kat.BD_INVALID)
    negative_offset_to(bwt, kat.BD_INVALID)
    check_predecessor_at(bwt)


@pytest.mark.parametrize("n,universe", [(1, 1), (5, 7), (31, 141), (1000, 100000), (1000, 1000), (3, 1 << 40)])
def test_sparse_select(n, universe):
    """Elias-Fano select/next against the plain sorted list (simple-sds semantics)."""
    import ctypes as C
    import random
    import numpy as np
    rng = random.Random(n * 31 + universe % 97)
    vals = sorted(rng.sample(range(universe), n)) if universe <= 10**6 else sorted(rng.randrange(universe) for _ in range(n))
    L = O.lib()
    sv = O.Sparse()
    arr = np.array(vals, dtype=np.uint64)
    assert L.go_sparse_build(C.byref(sv), universe, arr.ctypes.data, n) == 0
    pos = C.c_uint64(0)
    for i in range(n):
        assert L.go_sparse_select(C.byref(sv), i, C.byref(pos)) == vals[i]
        if i + 1 < n:
            assert L.go_sparse_next(C.byref(sv), i, C.byref(pos)) == vals[i + 1]
    L.go_sparse_free(C.byref(sv))
