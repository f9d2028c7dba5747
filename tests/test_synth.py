"""Host logic: simple-sds writer round trips, the GBWT builders, and the synthetic generator against the oracle."""
import os
import random

import numpy as np
import pytest

import kat
import oracle_lib as O
from gbwt_rs_amd import synth as S


@pytest.fixture(scope="module", autouse=True)
def _build():
    import subprocess
    subprocess.check_call(["make", "-C", os.path.dirname(S.LIB_PATH)], stdout=subprocess.DEVNULL)


@pytest.mark.parametrize("name,as_gbz", [("example.gbwt", False), ("with-empty.gbwt", False), ("translation.gbwt", False),
                                         ("example-v1.gbz", True), ("translation-v1.gbz", True)])
def test_writer_round_trip(tmp_path, name, as_gbz):
    """Loading a reference fixture with the product loader and writing it back reproduces the file byte for byte
    (Elias-Fano parameters, packed strings, dictionaries, headers, metadata, graph)."""
    src = os.path.join(O.GOLDEN, name)
    out = tmp_path / name
    S.Synth.from_file(src).save(str(out), as_gbz=as_gbz)
    assert out.read_bytes() == open(src, "rb").read()


def test_v2_gbz_rewritten_as_v1(tmp_path):
    """A zstd (v2) GBZ loads and is written back as a v1 container with the same content
    (the reference checks v1 / v2 equivalence the same way, src/gbz/tests.rs:572-612)."""
    for name in ("example.gbz", "translation.gbz"):
        out = tmp_path / ("v1-" + name)
        S.Synth.from_file(os.path.join(O.GOLDEN, name)).save(str(out), as_gbz=True)
        assert O.OracleGBZ(str(out)).gfa() == O.OracleGBZ(os.path.join(O.GOLDEN, name)).gfa()


@pytest.mark.parametrize("name,with_empty", [("example.gbwt", False), ("with-empty.gbwt", True)])
def test_builder_reproduces_fixture_records(name, with_empty):
    """The general builder rebuilds the fixtures' record streams byte for byte."""
    s = S.Synth.from_paths(kat.true_paths(with_empty), bidirectional=True)
    ref = O.OracleGBWT.load(os.path.join(O.GOLDEN, name))
    assert bytes(s.data()) == ref.bwt().data()
    assert list(s.starts()) == ref.bwt().starts()
    assert (s.sequences, s.size, s.alphabet_offset, s.alphabet_size) == (ref.sequences(), ref.len(), ref.alphabet_offset(), ref.alphabet_size())


def test_builder_reproduces_translation_records():
    paths = [[2 * x for x in p] for p in kat.TRANSLATION_PATHS]
    s = S.Synth.from_paths(paths, bidirectional=True)
    assert bytes(s.data()).hex() == kat.TRANSLATION_DATA_HEX
    assert list(s.starts()) == kat.TRANSLATION_STARTS


def oracle_of(s):
    bwt = O.OracleBWT.from_parts(bytes(s.data()), s.starts())
    return O.OracleGBWT.from_bwt(bwt, s.sequences, s.size, s.alphabet_offset, s.alphabet_size, s.bidirectional)


def random_paths(rng, n_paths, n_nodes, max_len, cyclic):
    paths = []
    for _ in range(n_paths):
        ln = rng.randint(0, max_len)
        if cyclic:
            p = [2 * rng.randint(1, n_nodes) + rng.randint(0, 1) for _ in range(ln)]
        else:
            ids = sorted(rng.sample(range(1, n_nodes + 1), min(ln, n_nodes)))
            p = [2 * i for i in ids]
        paths.append(p)
    return paths


@pytest.mark.parametrize("seed,cyclic", [(1, False), (2, True), (3, True), (4, False)])
def test_builder_random_paths_roundtrip(seed, cyclic):
    """Oracle extraction of a built index returns the paths that went in (both orientations)."""
    rng = random.Random(seed)
    paths = random_paths(rng, n_paths=12, n_nodes=9, max_len=14, cyclic=cyclic)
    if not any(paths):
        paths[0] = [2, 4]
    s = S.Synth.from_paths(paths, bidirectional=True)
    g = oracle_of(s)
    for i, p in enumerate(paths):
        assert g.sequence(2 * i) == p
        assert g.sequence(2 * i + 1) == kat.reverse_path(p)


@pytest.mark.parametrize("alleles,model", [(2, S.MOSAIC), (2, S.IID), (5, S.MOSAIC), (300, S.IID)])
def test_chain_generator_matches_bruteforce_builder(alleles, model):
    """The PBWT-sweep chain generator and the brute-force builder agree byte for byte."""
    s = S.Synth.chain(sites=7, haplotypes=23, alleles=alleles, model=model, founders=4, switch_rate=0.2, seed=5)
    paths = [[int(x) for x in s.path(h)] for h in range(s.paths)]
    b = S.Synth.from_paths(paths, bidirectional=True)
    assert bytes(s.data()) == bytes(b.data())
    assert list(s.starts()) == list(b.starts())
    assert (s.sequences, s.size, s.alphabet_offset, s.alphabet_size) == (b.sequences, b.size, b.alphabet_offset, b.alphabet_size)


@pytest.mark.parametrize("alleles,model,extra,every", [(2, S.MOSAIC, 1, 1), (2, S.IID, 3, 1), (5, S.MOSAIC, 2, 1), (40, S.IID, 1, 1),
                                                       (2, S.MOSAIC, 2, 3), (4, S.IID, 1, 4), (2, S.IID, 1, 9)])
def test_indel_chain_matches_bruteforce_builder(alleles, model, extra, every):
    """Insertion alleles (paths of different lengths): the sweep generator still agrees with the brute-force builder
    byte for byte, and its checksums and header follow the longer paths."""
    s = S.Synth.chain(sites=9, haplotypes=21, alleles=alleles, model=model, founders=4, switch_rate=0.2, seed=8, extra=extra, indel_every=every)
    paths = [[int(x) for x in s.path(h)] for h in range(s.paths)]
    assert len({len(p) for p in paths}) > 1
    for h, p in enumerate(paths):
        assert sum(p) == s.path_checksum(h)
    b = S.Synth.from_paths(paths, bidirectional=True)
    assert bytes(s.data()) == bytes(b.data())
    assert list(s.starts()) == list(b.starts())
    assert (s.sequences, s.size, s.alphabet_offset, s.alphabet_size) == (b.sequences, b.size, b.alphabet_offset, b.alphabet_size)


@pytest.mark.parametrize("alleles,model,extra,every,chop", [(2, S.MOSAIC, 0, 1, 2), (2, S.IID, 1, 1, 3), (3, S.MOSAIC, 2, 2, 2), (5, S.IID, 0, 1, 4),
                                                            (2, S.MOSAIC, 1, 5, 5)])
def test_chopped_chain_matches_bruteforce_builder(alleles, model, extra, every, chop):
    """Every node a chain of `chop` nodes with consecutive ids (most records unary, as in a GBZ built from a GFA with long segments):
    byte for byte the brute-force builder's index."""
    s = S.Synth.chain(sites=7, haplotypes=19, alleles=alleles, model=model, founders=4, switch_rate=0.2, seed=12, extra=extra, indel_every=every, chop=chop)
    paths = [[int(x) for x in s.path(h)] for h in range(s.paths)]
    for h, p in enumerate(paths):
        assert sum(p) == s.path_checksum(h)
        assert all(b - a == 2 for a, b in zip(p[:chop], p[1:chop]))           # the anchor's pieces: consecutive ids, forward
        # ... and so are the pieces of every allele (a chopped GFA segment): a path is made of runs of consecutive ids, one per logical
        # node -- the anchor (chop pieces), then the allele (chop, or chop * (1 + extra) where it is an insertion) -- or longer where the
        # ids of an allele happen to run on into the next anchor
        runs, k = [], 0
        while k < len(p):
            j = k + 1
            while j < len(p) and p[j] - p[j - 1] == 2:
                j += 1
            runs.append(j - k)
            k = j
        assert all(r % chop == 0 for r in runs) and len(runs) <= 2 * 7
    b = S.Synth.from_paths(paths, bidirectional=True)
    assert bytes(s.data()) == bytes(b.data())
    assert list(s.starts()) == list(b.starts())
    assert (s.sequences, s.size, s.alphabet_offset, s.alphabet_size) == (b.sequences, b.size, b.alphabet_offset, b.alphabet_size)


def test_indel_chain_gbz_loads_in_the_oracle(tmp_path):
    s = S.Synth.chain(sites=150, haplotypes=40, alleles=2, model=S.MOSAIC, founders=6, switch_rate=0.05, seed=3, extra=2)
    path = tmp_path / "indel.gbz"
    s.save(str(path), as_gbz=True)
    z = O.OracleGBZ(str(path))
    g = z.gbwt()
    for h in (0, 7, 39):
        truth = [int(x) for x in s.path(h)]
        assert g.sequence(2 * h) == truth
        assert g.sequence(2 * h + 1) == kat.reverse_path(truth)
    w = [l for l in z.gfa().split(b"\n") if l.startswith(b"W\t")]
    assert len(w) == 39 and int(w[0].split(b"\t")[5]) == len(s.path(1))   # 1 bp labels: walk length in bases = nodes


def test_chopped_chain_gbz_loads_in_the_oracle(tmp_path):
    s = S.Synth.chain(sites=60, haplotypes=30, alleles=2, model=S.MOSAIC, founders=6, switch_rate=0.05, seed=4, extra=1, indel_every=3, chop=3)
    path = tmp_path / "chopped.gbz"
    s.save(str(path), as_gbz=True)
    z = O.OracleGBZ(str(path))
    g = z.gbwt()
    for h in (0, 11, 29):
        truth = [int(x) for x in s.path(h)]
        assert g.sequence(2 * h) == truth
        assert g.sequence(2 * h + 1) == kat.reverse_path(truth)
    w = [l for l in z.gfa().split(b"\n") if l.startswith(b"W\t")]
    assert len(w) == 29 and int(w[0].split(b"\t")[5]) == len(s.path(1))   # 1 bp labels on every piece


def test_chain_generator_oracle_extraction(tmp_path):
    """Config C2 shape at reduced size: file -> oracle loader -> extraction == generator ground truth; GFA lines parse."""
    s = S.Synth.chain(sites=200, haplotypes=64, alleles=2, model=S.MOSAIC, founders=8, switch_rate=0.02, seed=42)
    path = tmp_path / "chain.gbz"
    s.save(str(path), as_gbz=True)
    z = O.OracleGBZ(str(path))
    g = z.gbwt()
    assert g.sequences() == 128 and g.len() == s.size
    offsets, nodes = g.extract(list(range(0, 128, 2)), threads=2)
    for h in range(64):
        truth = s.path(h)
        assert np.array_equal(nodes[offsets[h]:offsets[h + 1]], truth)
        assert int(truth.astype(np.uint64).sum()) == s.path_checksum(h)
    rev = g.sequence(2 * 5 + 1)
    assert rev == kat.reverse_path([int(x) for x in s.path(5)])
    gfa = z.gfa().split(b"\n")
    assert gfa[0] == b"H\tVN:Z:1.1"
    p_lines = [l for l in gfa if l.startswith(b"P\t")]
    w_lines = [l for l in gfa if l.startswith(b"W\t")]
    assert len(p_lines) == 1 and len(w_lines) == 63
    assert p_lines[0].startswith(b"P\tchr1\t1+,")
    f = w_lines[0].split(b"\t")
    assert f[1:6] == [b"s0", b"1", b"chr1", b"0", b"400"]


def test_chain_high_degree_uses_two_varint_runs():
    """Config C5 shape: with >= 255 alleles in use the anchor records switch to the (value, len - 1) varint code."""
    s = S.Synth.chain(sites=3, haplotypes=3000, alleles=300, model=S.IID, zipf=0.2, seed=9)
    g = oracle_of(s)
    rec = g.bwt().record(1)   # first anchor, forward
    assert rec.outdegree >= 255
    for h in (0, 17, 2999):
        assert g.sequence(2 * h) == [int(x) for x in s.path(h)]


def test_merged_genome_matches_the_brute_force_builder(tmp_path):
    """Synth.genome (config C4's shape: contigs x fragments = graph components, ragged walks of several samples and phases with
    non-zero fragment offsets, one generic path per contig): the merged record stream is what the reverse-prefix sort builds from
    the same paths, byte for byte, and the oracle reads the saved GBZ back with the metadata that went in."""
    g = S.Synth.genome(contigs=6, fragments=3, haplotypes=10, sites=24, seed=5)
    paths = [[int(x) for x in g.path(p)] for p in range(g.paths)]
    b = S.Synth.from_paths(paths, bidirectional=True)
    assert bytes(b.data()) == bytes(g.data()) and np.array_equal(b.starts(), g.starts())
    assert (b.size, b.sequences, b.alphabet_size, b.alphabet_offset) == (g.size, g.sequences, g.alphabet_size, g.alphabet_offset)
    assert all(g.path_checksum(p) == sum(paths[p]) for p in range(g.paths))
    assert len({len(p) for p in paths}) > 10                                   # ragged
    path = tmp_path / "genome.gbz"
    g.save(str(path), as_gbz=True)
    z = O.OracleGBZ(str(path))
    assert z.paths() == g.paths
    for p in range(0, g.paths, 7):
        assert [2 * n + o for n, o in z.path(p)] == paths[p]
    text = z.gfa()
    p_lines = [l for l in text.split(b"\n") if l.startswith(b"P\t")]
    w_lines = [l for l in text.split(b"\n") if l.startswith(b"W\t")]
    assert sorted(l.split(b"\t")[1] for l in p_lines) == sorted(f"chr{c + 1}".encode() for c in range(6))     # one generic path per contig
    assert len(p_lines) + len(w_lines) == g.paths
    fields = [l.split(b"\t") for l in w_lines]
    assert len({f[3] for f in fields}) == 6 and len({f[1] for f in fields}) == 5 and {f[2] for f in fields} == {b"1", b"2"}
    assert sum(int(f[4]) != 0 for f in fields) > len(fields) // 2               # fragment offsets in use
    assert all(int(f[5]) > int(f[4]) for f in fields)                            # end = fragment + length (src/bin/gbunzip.rs:508-519)
    assert z.pan_sn_path(1).count("#") == 2
