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
