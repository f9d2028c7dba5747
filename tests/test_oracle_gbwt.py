"""Oracle GBWT navigation/search on the reference fixtures; mirrors src/gbwt/tests.rs:44-462."""
import os

import numpy as np
import pytest

import kat
import oracle_lib as O

FIXTURES = [("example.gbwt", False), ("with-empty.gbwt", True)]


def load(name):
    return O.OracleGBWT.load(os.path.join(O.GOLDEN, name))


def extract_sequence(index, i):  # src/gbwt/tests.rs:92-100
    out, pos = [], index.start(i)
    while pos is not None:
        out.append(pos[0])
        pos = index.forward(pos)
    return out


def test_statistics():  # src/gbwt/tests.rs:44-56
    for name, stats in [("example.gbwt", (68, 12, 52, 21)), ("with-empty.gbwt", (70, 14, 52, 21))]:
        g = load(name)
        assert (g.len(), g.sequences(), g.alphabet_size(), g.alphabet_offset()) == stats
        assert g.is_bidirectional() and g.first_node() == 22
    assert load("example.gbwt").has_metadata()
    assert not load("with-empty.gbwt").has_metadata()


def test_raw_streams():
    """Record byte stream + record starts of the fixtures equal SURVEY Appendix B."""
    bwt = load("example.gbwt").bwt()
    assert bwt.data().hex() == kat.EXAMPLE_DATA_HEX and len(bwt.data()) == 141
    assert bwt.starts() == kat.EXAMPLE_STARTS
    z = O.OracleGBZ(os.path.join(O.GOLDEN, "translation.gbz"))
    tb = z.gbwt().bwt()
    assert tb.data().hex() == kat.TRANSLATION_DATA_HEX and tb.starts() == kat.TRANSLATION_STARTS


@pytest.mark.parametrize("name,with_empty", FIXTURES)
def test_extract(name, with_empty):  # src/gbwt/tests.rs:164-189
    g = load(name)
    truth = kat.true_paths(with_empty)
    assert g.sequences() // 2 == len(truth)
    for i in range(g.sequences() // 2):
        f = extract_sequence(g, 2 * i)
        assert f == truth[i]
        assert extract_sequence(g, 2 * i + 1) == kat.reverse_path(f)


def test_example_sequences_appendix_b():
    g = load("example.gbwt")
    for i, seq in kat.EXAMPLE_SEQUENCES.items():
        assert g.sequence(i) == seq
    assert g.sequence(7) == [35, 33, 29, 27, 23]   # doc-test src/gbwt.rs:546-548


@pytest.mark.parametrize("name,with_empty", FIXTURES)
def test_backward(name, with_empty):  # src/gbwt/tests.rs:191-214
    g = load(name)
    for i in range(g.sequences()):
        f = extract_sequence(g, i)
        last, pos = None, g.start(i)
        while pos is not None:
            last, pos = pos, g.forward(pos)
        r, pos = [], last
        while pos is not None:
            r.append(pos[0])
            pos = g.backward(pos)
        assert r == list(reversed(f))


@pytest.mark.parametrize("name,with_empty", FIXTURES)
def test_sequence(name, with_empty):  # src/gbwt/tests.rs:216-238
    g = load(name)
    for i in range(g.sequences()):
        assert g.sequence(i) == extract_sequence(g, i)
    assert g.sequence(g.sequences()) is None
    if with_empty:
        assert g.sequence(8) == [] and g.sequence(9) == [] and g.start(8) is None


@pytest.mark.parametrize("name,with_empty", FIXTURES)
def test_find(name, with_empty):  # src/gbwt/tests.rs:268-292
    g = load(name)
    nodes = kat.true_nodes()
    for i in range(g.alphabet_size() + 1):
        st = g.find(i)
        if st is not None:
            assert i in nodes and st[0] == i and st[2] > st[1]
        else:
            assert i not in nodes


@pytest.mark.parametrize("name,with_empty", FIXTURES)
def test_extend(name, with_empty):  # src/gbwt/tests.rs:294-350
    g = load(name)
    paths = kat.true_paths(with_empty)
    for first in kat.true_nodes():
        start = g.find(first)
        for i in range(g.alphabet_size() + 1):
            count = kat.count_occurrences(paths, [first, i])
            st = g.extend(start, i)
            assert (st[2] - st[1] if st else 0) == count
    for path in paths:
        for j in range(len(path)):
            fw = g.find(path[j])
            for k in range(j + 1, len(path)):
                fw = g.extend(fw, path[k])
                assert fw is not None and fw[2] - fw[1] == kat.count_occurrences(paths, path[j:k + 1])
            bw = g.find(kat.flip(path[j]))
            for k in range(j - 1, -1, -1):
                bw = g.extend(bw, kat.flip(path[k]))
                assert bw is not None and bw[2] - bw[1] == kat.count_occurrences(paths, path[k:j + 1])


def bd_search(g, path, first, start, end):  # src/gbwt/tests.rs:354-363
    st = g.bd_find(path[first])
    if st is None:
        return None
    for i in range(first + 1, end):
        st = g.extend_forward(st, path[i])
        if st is None:
            return None
    for i in range(first - 1, start - 1, -1):
        st = g.extend_backward(st, path[i])
        if st is None:
            return None
    return st


@pytest.mark.parametrize("name,with_empty", FIXTURES)
def test_bd_find(name, with_empty):  # src/gbwt/tests.rs:365-391
    g = load(name)
    nodes = kat.true_nodes()
    for i in range(g.alphabet_size() + 1):
        st = g.bd_find(i)
        if st is not None:
            assert i in nodes and st[0][0] == i and st[1][0] == kat.flip(i)
            assert st[0][2] - st[0][1] == st[1][2] - st[1][1] > 0
        else:
            assert i not in nodes


@pytest.mark.parametrize("name,with_empty", FIXTURES)
def test_bd_extend(name, with_empty):  # src/gbwt/tests.rs:393-462
    g = load(name)
    paths = kat.true_paths(with_empty)
    for first in kat.true_nodes():
        start = g.bd_find(first)
        for i in range(g.alphabet_size() + 1):
            st = g.extend_forward(start, i)
            assert (st[0][2] - st[0][1] if st else 0) == kat.count_occurrences(paths, [first, i])
            st = g.extend_backward(start, i)
            assert (st[0][2] - st[0][1] if st else 0) == kat.count_occurrences(paths, [i, first])
    for path in paths:
        for p in (path, kat.reverse_path(path)):
            for first in range(len(p)):
                for s in range(first + 1):
                    for e in range(first + 1, len(p) + 1):
                        st = bd_search(g, p, first, s, e)
                        assert st is not None
                        n = st[0][2] - st[0][1]
                        assert n == kat.count_occurrences(paths, p[s:e]) == st[1][2] - st[1][1]
                        assert st[0][0] == p[e - 1] and st[1][0] == kat.flip(p[s])


def test_doc_search_states():
    # doc-test src/gbwt.rs:70-83
    g = load("example.gbwt")
    st = g.find(24)
    st = g.extend(st, 28)
    st = g.extend(st, 30)
    assert st[0] == 30 and st[2] - st[1] == 2
    bd = g.bd_find(28)
    bd = g.extend_backward(bd, 24)
    bd = g.extend_forward(bd, 30)
    assert bd == ((30, 0, 2), (25, 0, 2))


def test_doc_state_iter():
    # doc-test of StateIter, src/gbz.rs:1184-1209: node 14 forward has 3 paths and 2 successors; the predecessors of the
    # successors are (12 -> 15, 2 paths) and (13 -> 16, 1 path)
    g = load("example.gbwt")
    state = g.bd_find(2 * 14)
    assert state[0][0] == 28 and state[0][2] - state[0][1] == 3
    successors = g.follow(state)
    assert len(successors) == 2
    preds = [p for s in successors for p in g.follow(s, backward=True)]
    summary = [(p[1][0] ^ 1, p[0][0], p[0][2] - p[0][1]) for p in preds]     # (from, to, len)
    assert summary == [(2 * 12, 2 * 15, 2), (2 * 13, 2 * 16, 1)]
    assert g.follow(((0, 0, 1), (1, 0, 1))) is None


def test_extract_batched_and_bytes():
    g = load("example.gbwt")
    ids = list(range(12))
    for threads in (1, 3):
        offsets, nodes = g.extract(ids, threads=threads)
        assert int(offsets[-1]) == 68 - 12   # size - sequences (src/gbwt.rs:108-122)
        for i in ids:
            assert list(nodes[offsets[i]:offsets[i + 1]]) == kat.EXAMPLE_SEQUENCES[i]
    total, steps = g.algorithmic_bytes(ids)
    assert steps == 56 and total > 4 * 56


def test_rejects_bad_files(tmp_path):
    raw = bytearray(open(os.path.join(O.GOLDEN, "example.gbwt"), "rb").read())
    bad = bytearray(raw); bad[0] ^= 0xFF                      # tag (src/headers.rs:102-104)
    p = tmp_path / "bad-tag.gbwt"; p.write_bytes(bad)
    with pytest.raises(ValueError, match="Invalid tag"):
        O.OracleGBWT.load(str(p))
    bad = bytearray(raw); bad[4] = 4                          # version (src/headers.rs:105-114)
    p = tmp_path / "bad-version.gbwt"; p.write_bytes(bad)
    with pytest.raises(ValueError, match="Invalid version"):
        O.OracleGBWT.load(str(p))
    bad = bytearray(raw); bad[40] = 0x0F                      # unknown flag bit
    p = tmp_path / "bad-flags.gbwt"; p.write_bytes(bad)
    with pytest.raises(ValueError, match="Invalid flags"):
        O.OracleGBWT.load(str(p))
    bad = bytearray(raw); bad[40] = 0x03                      # simple-sds flag missing (src/headers.rs:229-231)
    p = tmp_path / "sdsl.gbwt"; p.write_bytes(bad)
    with pytest.raises(ValueError, match="SDSL"):
        O.OracleGBWT.load(str(p))
    bad = bytearray(raw); bad[42 * 8] = 140                   # data length != index universe (src/bwt.rs:179-181)
    p = tmp_path / "mismatch.gbwt"; p.write_bytes(bad)
    with pytest.raises(ValueError):
        O.OracleGBWT.load(str(p))
    with pytest.raises(ValueError, match="not bidirectional|Invalid tag"):
        O.OracleGBZ(os.path.join(O.GOLDEN, "example.gbwt"))


def test_batched_search_matches_single_calls():
    g = load("example.gbwt")
    paths = kat.true_paths(False)
    queries = [p[j:j + 3] for p in paths + [kat.reverse_path(x) for x in paths] for j in range(len(p) - 2)] + [[24, 30, 34], [0, 22, 24]]
    out, ok = g.search_batch(queries, threads=3)
    for q, r, v in zip(queries, out, ok):
        exp = g.find(q[0])
        for x in q[1:]:
            exp = g.extend(exp, x) if exp else None
        assert bool(v) == (exp is not None)
        if exp:
            assert tuple(int(x) for x in r) == exp
    for first in (0, 1, 2):
        bout, bok = g.bd_search_batch(queries, first, threads=2)
        for q, r, v in zip(queries, bout, bok):
            exp = bd_search(g, q, first, 0, 3)
            # bd_search() extends forward first, then backward; the batched form alternates -- same final state
            assert bool(v) == (exp is not None)
            if exp:
                assert (tuple(int(x) for x in r[:3]), tuple(int(x) for x in r[3:])) == exp


def _splitmix64(i):
    m = (1 << 64) - 1
    z = (i + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def test_extract_checksums_follow_the_header_definition():
    """go_gbwt_extract_sums_mt (what bench.py's cpu_baseline leg times and compares with the GPU): per sequence its length, the sum of its
    node ids and the order-dependent hash of include/gbwt_hip.h (gbwt_hip_path_hashes), sum of (node + 1) * splitmix64(position) -- checked
    here against the known paths of the fixtures (src/gbwt/tests.rs:44-49) with the definition written out in Python."""
    for name in ("example.gbwt", "with-empty.gbwt"):
        oracle = O.OracleGBWT.load(os.path.join(O.GOLDEN, name))
        ids = np.arange(oracle.sequences() + 2, dtype=np.uint64)          # two ids without a sequence: empty rows
        for threads in (1, 3):
            steps, lengths, sums, hashes = oracle.extract_checksums(ids, threads)
            off, nodes = oracle.extract(ids, threads)
            assert steps == int(off[-1])
            for k in range(len(ids)):
                row = [int(v) for v in nodes[int(off[k]):int(off[k + 1])]]
                assert int(lengths[k]) == len(row) and int(sums[k]) == sum(row)
                assert int(hashes[k]) == sum((v + 1) * _splitmix64(i) for i, v in enumerate(row)) % (1 << 64)
    # the hash tells two orders of the same nodes apart, the sum does not
    assert (23 * _splitmix64(0) + 25 * _splitmix64(1)) % (1 << 64) != (25 * _splitmix64(0) + 23 * _splitmix64(1)) % (1 << 64)
