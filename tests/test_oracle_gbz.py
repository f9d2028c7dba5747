"""Oracle GBZ paths and gbunzip GFA text on the fixtures; mirrors src/gbz/tests.rs:85-98,279-292,371-381,572-612."""
import hashlib
import os

import pytest

import kat
import oracle_lib as O


def gbz(name):
    return O.OracleGBZ(os.path.join(O.GOLDEN, name))


@pytest.mark.parametrize("name", ["example.gbz", "example-v1.gbz"])
def test_example_paths(name):
    z = gbz(name)
    truth = kat.true_paths(False)
    assert z.paths() == len(truth)
    for i, t in enumerate(truth):
        assert z.path(i) == [(x // 2, x & 1) for x in t]
        assert z.path(i, reverse=True) == [(x // 2, x & 1) for x in kat.reverse_path(t)]
    assert z.path(z.paths()) is None


@pytest.mark.parametrize("name", ["translation.gbz", "translation-v1.gbz"])
def test_translation_paths(name):
    z = gbz(name)
    assert z.paths() == 3
    for i, t in enumerate(kat.TRANSLATION_PATHS):
        assert z.path(i) == [(x, 0) for x in t]
        assert z.path(i, reverse=True) == [(x, 1) for x in reversed(t)]


@pytest.mark.parametrize("name", ["example.gbz", "example-v1.gbz"])
def test_example_gfa(name):
    """Config C1 plumbing: example.gbz -> the 454 bytes of SURVEY Appendix C."""
    text = gbz(name).gfa()
    assert len(text) == kat.EXAMPLE_GFA_LEN
    assert hashlib.sha256(text).hexdigest() == kat.EXAMPLE_GFA_SHA256
    assert text.endswith(kat.EXAMPLE_PW_LINES)
    # equals test-data/example.gfa except the header version and the P-line overlap column
    ref = open(os.path.join(O.GOLDEN, "example.gfa"), "rb").read().split(b"\n")
    got = text.split(b"\n")
    assert len(ref) == len(got)
    for r, g in zip(ref, got):
        if r.startswith(b"H"):
            assert (r, g) == (b"H\tVN:Z:1.0", b"H\tVN:Z:1.1")
        elif r.startswith(b"P"):
            assert r.split(b"\t")[:3] == g.split(b"\t")[:3] and g.split(b"\t")[3] == b"*"
        else:
            assert r == g


@pytest.mark.parametrize("name", ["translation.gbz", "translation-v1.gbz"])
def test_translation_gfa(name):
    text = gbz(name).gfa()
    assert len(text) == kat.TRANSLATION_GFA_LEN
    assert hashlib.sha256(text).hexdigest() == kat.TRANSLATION_GFA_SHA256
    assert b"W\tsample\t1\tA\t0\t10\t>s11>s12>s14>s15>s17\n" in text


def test_path_lines_subset():
    z = gbz("example.gbz")
    assert z.path_lines([0, 1], 0) + z.path_lines([2, 3, 4, 5], 1) == kat.EXAMPLE_PW_LINES


def test_pan_sn_names():
    """Metadata::pan_sn_path, doc-test src/gbwt.rs:596-598: path 3 of the example is "sample#2#A"; generic paths carry phase 0
    in memory (src/gbwt.rs:879-886)."""
    z = gbz("example.gbz")
    assert z.pan_sn_path(3) == "sample#2#A"
    assert [z.pan_sn_path(i) for i in range(6)] == ["_gbwt_ref#0#A", "_gbwt_ref#0#B", "sample#1#A", "sample#2#A", "sample#1#B", "sample#2#B"]
    assert z.pan_sn_path(6) is None


@pytest.mark.parametrize("name", ["example.gbz", "translation.gbz"])
def test_path_modes(name):
    """gbunzip --paths default / pan-sn / ref-only (src/bin/gbunzip.rs:63-76, 212-222): the three files share H, S and L lines;
    pan-sn writes every path as a P-line named sample#phase#contig (write_pan_sn 371-393), ref-only stops after the P-lines
    of the generic sample."""
    z = gbz(name)
    default, pan_sn, ref_only = z.gfa(0), z.gfa(1), z.gfa(2)
    assert default == z.gfa()
    split = default.index(b"P\t")
    assert pan_sn[:split] == default[:split] and ref_only[:split] == default[:split]
    n = z.paths()
    assert pan_sn[split:] == z.path_lines(list(range(n)), 2)
    generic = [i for i in range(n) if z.pan_sn_path(i).startswith("_gbwt_ref#")]
    assert ref_only[split:] == z.path_lines(generic, 0)
    assert default == ref_only + z.path_lines([i for i in range(n) if i not in generic], 1)
    # a PanSN line is the P-line of the path under another name
    for i in range(n):
        p, q = z.path_lines([i], 0).split(b"\t"), z.path_lines([i], 2).split(b"\t")
        assert q[1] == z.pan_sn_path(i).encode() and p[0] == q[0] == b"P" and p[2:] == q[2:]
    if name == "example.gbz":
        assert z.path_lines([3], 2) == b"P\tsample#2#A\t11+,13+,14+,16+,17+\t*\n"


def test_segment_paths_known_answers():
    """src/gbz/tests.rs:466-497 (segment_paths): translation.gbz has three paths as sequences of segment identifiers; the reverse orientation is
    the reversed list with every orientation flipped; a past-the-end path and a graph without a translation give None.  The doc-test of
    SegmentPathIter (src/gbz.rs:1080-1092): path 2 reversed = s17-, s16-, s14-, s13-, s11- (segment ids 7, 6, 3, 2, 0)."""
    z = O.OracleGBZ(os.path.join(O.GOLDEN, "translation.gbz"))
    truth = [[0, 1, 3, 5, 7], [0, 1, 3, 5, 7], [0, 2, 3, 6, 7]]
    for p, segments in enumerate(truth):
        assert z.segment_path(2 * p) == [(s, 0) for s in segments]
        assert z.segment_path(2 * p + 1) == [(s, 1) for s in reversed(segments)]
    assert z.segment_path(2 * 2 + 1) == [(7, 1), (6, 1), (3, 1), (2, 1), (0, 1)]
    assert z.segment_path(2 * len(truth)) is None and z.segment_path(2 * len(truth) + 1) is None
    assert O.OracleGBZ(os.path.join(O.GOLDEN, "example.gbz")).segment_path(0) is None            # no translation
