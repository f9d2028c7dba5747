"""GFA text through the product path (device walk + device token formatting) against the oracle's gbunzip
restatement and the expected bytes of SURVEY Appendix C (config C1: example.gbz -> GFA)."""
import hashlib
import os

import numpy as np
import pytest

import gbwt_rs_amd as G
import kat
import oracle_lib as O
from gbwt_rs_amd import synth as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["example.gbz", "example-v1.gbz"])
def test_example_gbz_to_gfa(tmp_path, name):
    path = os.path.join(O.GOLDEN, name)
    dev = G.GBZ.load(path)
    out = tmp_path / "out.gfa"
    dev.write_gfa(str(out))
    text = out.read_bytes()
    assert len(text) == kat.EXAMPLE_GFA_LEN
    assert hashlib.sha256(text).hexdigest() == kat.EXAMPLE_GFA_SHA256
    assert text == O.OracleGBZ(path).gfa()
    assert dev.path_lines([0, 1], 0) + dev.path_lines([2, 3, 4, 5], 1) == kat.EXAMPLE_PW_LINES
    assert dev.path_lines([5, 2], 1) == O.OracleGBZ(path).path_lines([5, 2], 1)
    assert dev.path_lines([], 1) == b""
    empty = dev.path_lines_device([], 1)
    assert empty.total == 0 and empty.n == 0
    lines = dev.path_lines_device([5, 2], 1)                 # device-resident: the copy-out of the same request finds it there
    assert lines.n == 2 and lines.total == len(O.OracleGBZ(path).path_lines([5, 2], 1))
    assert dev.path_lines([5, 2], 1) == O.OracleGBZ(path).path_lines([5, 2], 1)


@pytest.mark.parametrize("name", ["translation.gbz", "translation-v1.gbz"])
def test_translation_gbz_to_gfa(tmp_path, name):
    """Segment names instead of node ids (SegmentPathIter, src/gbz.rs:1098-1169)."""
    path = os.path.join(O.GOLDEN, name)
    dev, oracle = G.GBZ.load(path), O.OracleGBZ(path)
    out = tmp_path / "out.gfa"
    dev.write_gfa(str(out))
    text = out.read_bytes()
    assert len(text) == kat.TRANSLATION_GFA_LEN
    assert hashlib.sha256(text).hexdigest() == kat.TRANSLATION_GFA_SHA256
    assert text == oracle.gfa()
    ids = list(range(dev.paths()))
    for mode in (0, 1):
        assert dev.path_lines(ids, mode) == oracle.path_lines(ids, mode)
        assert dev.path_lines(ids[::-1], mode) == oracle.path_lines(ids[::-1], mode)


def translated_graph(tmp_path, paths, segment_starts, name):
    s = S.Synth.from_paths(paths, bidirectional=True).attach_gbz(segment_starts, seed=5)
    path = tmp_path / name
    s.save(str(path), as_gbz=True)
    return G.GBZ.load(str(path)), O.OracleGBZ(str(path))


def test_synthetic_translation_valid_paths(tmp_path):
    """Multi-node segments walked in both orientations, paths longer than one formatting chunk."""
    # segments: [1,2,3] [4] [5,6] [7] [8,9,10,11] [12]
    starts = [1, 4, 5, 7, 8, 12]
    fwd = lambda a, b: [2 * v for v in range(a, b + 1)]
    rev = lambda a, b: [2 * v + 1 for v in range(b, a - 1, -1)]
    unit = fwd(1, 3) + fwd(4, 4) + rev(5, 6) + fwd(7, 7) + fwd(8, 11) + rev(12, 12)
    paths = [unit, fwd(1, 3) + fwd(5, 6) + fwd(8, 11), rev(8, 11) + fwd(4, 4) + rev(1, 3), unit * 60, [], fwd(12, 12), unit * 800]   # the last one: three chunks of 4 096 positions, segments straddle them
    dev, oracle = translated_graph(tmp_path, paths, starts, "segments.gbz")
    out = tmp_path / "segments.gfa"
    dev.write_gfa(str(out))
    assert out.read_bytes() == oracle.gfa()
    ids = list(range(len(paths)))
    for mode in (0, 1):
        assert dev.path_lines(ids, mode) == oracle.path_lines(ids, mode)


def test_synthetic_translation_broken_paths(tmp_path):
    """Paths that are not concatenations of whole segments: the reference's iterator stops inside them (fail flag,
    src/gbz.rs:1147-1166); lines must be cut exactly where it stops."""
    starts = [1, 4, 5, 7, 8, 12]
    fwd = lambda a, b: [2 * v for v in range(a, b + 1)]
    rev = lambda a, b: [2 * v + 1 for v in range(b, a - 1, -1)]
    paths = [fwd(1, 3) + fwd(4, 4),                # valid
             fwd(2, 3) + fwd(4, 4),                # enters segment [1,2,3] at its second node
             fwd(1, 2) + fwd(4, 4),                # leaves segment [1,2,3] early
             fwd(1, 3) + rev(5, 5) + fwd(7, 7),    # enters [5,6] reversed at the wrong end
             fwd(4, 4) + fwd(9, 9),                # ends inside a segment entered in the middle
             fwd(8, 9) + rev(9, 9),                # turns around inside a segment
             fwd(1, 3) + fwd(1, 3),                # valid: the same segment twice
             fwd(12, 12)]                          # (makes node 12 exist)
    dev, oracle = translated_graph(tmp_path, paths, starts, "broken.gbz")
    ids = list(range(len(paths)))
    for mode in (0, 1):
        for i in ids:
            assert dev.path_lines([i], mode) == oracle.path_lines([i], mode), (i, mode)
        assert dev.path_lines(ids, mode) == oracle.path_lines(ids, mode)
    out = tmp_path / "broken.gfa"
    dev.write_gfa(str(out))
    assert out.read_bytes() == oracle.gfa()


@pytest.mark.parametrize("name", ["example.gbz", "example-v1.gbz", "translation.gbz"])
def test_path_modes_on_fixtures(tmp_path, name):
    """gbunzip --paths default / pan-sn / ref-only (PathMode, src/bin/gbunzip.rs:63-76, 212-222) through gbwt_hip_write_gfa_mode, and
    the PanSN P-lines (mode 2 of gbwt_hip_path_lines; path_to_pan_sn 487-491, Metadata::pan_sn_path src/gbwt.rs:709-713)."""
    path = os.path.join(O.GOLDEN, name)
    dev, oracle = G.GBZ.load(path), O.OracleGBZ(path)
    for mode in (G.PATHS_DEFAULT, G.PATHS_PAN_SN, G.PATHS_REF_ONLY):
        out = tmp_path / f"mode{mode}.gfa"
        dev.write_gfa(str(out), mode)
        assert out.read_bytes() == oracle.gfa(mode), mode
    ids = list(range(dev.paths()))
    assert dev.path_lines(ids, 2) == oracle.path_lines(ids, 2)
    assert dev.path_lines(ids[::-1], 2) == oracle.path_lines(ids[::-1], 2)
    if name.startswith("example"):
        assert dev.path_lines([3], 2) == b"P\tsample#2#A\t11+,13+,14+,16+,17+\t*\n"      # pan_sn_path(3) == "sample#2#A", src/gbwt.rs:598
    with pytest.raises(G.GbwtHipError):
        dev.path_lines(ids, 3)
    with pytest.raises(G.GbwtHipError):
        dev.write_gfa(str(tmp_path / "bad.gfa"), 3)


def test_path_modes_on_synthetic(tmp_path):
    """The three path modes on a generated GBZ with hundreds of haplotypes and on a translated graph with broken segment paths."""
    s = S.Synth.chain(sites=500, haplotypes=260, alleles=3, model=S.MOSAIC, founders=8, switch_rate=0.02, seed=19)
    path = tmp_path / "synth.gbz"
    s.save(str(path), as_gbz=True)
    dev, oracle = G.GBZ.load(str(path)), O.OracleGBZ(str(path))
    for mode in (G.PATHS_DEFAULT, G.PATHS_PAN_SN, G.PATHS_REF_ONLY):
        out = tmp_path / f"synth{mode}.gfa"
        dev.write_gfa(str(out), mode)
        got, exp = out.read_bytes(), oracle.gfa(mode)
        assert hashlib.sha256(got).hexdigest() == hashlib.sha256(exp).hexdigest() and got == exp, mode
    starts = [1, 4, 5, 7, 8, 12]
    fwd = lambda a, b: [2 * v for v in range(a, b + 1)]
    rev = lambda a, b: [2 * v + 1 for v in range(b, a - 1, -1)]
    paths = [fwd(1, 3) + fwd(4, 4), fwd(2, 3) + fwd(4, 4), fwd(1, 3) + rev(5, 6) + fwd(7, 7), fwd(8, 9) + rev(9, 9), (fwd(1, 3) + rev(5, 6)) * 2000, fwd(12, 12)]
    dev, oracle = translated_graph(tmp_path, paths, starts, "modes.gbz")
    for mode in (G.PATHS_DEFAULT, G.PATHS_PAN_SN, G.PATHS_REF_ONLY):
        out = tmp_path / f"modes{mode}.gfa"
        dev.write_gfa(str(out), mode)
        assert out.read_bytes() == oracle.gfa(mode), mode
    ids = list(range(len(paths)))
    assert dev.path_lines(ids, 2) == oracle.path_lines(ids, 2)


def test_segment_paths_as_data(tmp_path):
    """gbwt_hip_segment_paths = GBZ::segment_path for a batch of sequence ids (src/gbz.rs:477-489; SegmentPathIter 1098-1169): the reference's own
    known answers on translation.gbz (src/gbz/tests.rs:466-497: three paths as segment ids, reverse = reversed and flipped; past-the-end ids are
    empty rows), valid paths of several chunks and broken paths (the iterator stops inside them) against the oracle in both orientations, in one
    batch and one by one; a graph without a translation: BAD_ARGUMENT (the reference returns None)."""
    dev, oracle = G.GBZ.load(os.path.join(O.GOLDEN, "translation.gbz")), O.OracleGBZ(os.path.join(O.GOLDEN, "translation.gbz"))
    truth = [[0, 1, 3, 5, 7], [0, 1, 3, 5, 7], [0, 2, 3, 6, 7]]
    for p, segments in enumerate(truth):
        assert dev.segment_path(p, G.FORWARD) == [(s, 0) for s in segments] == oracle.segment_path(2 * p)
        assert dev.segment_path(p, G.REVERSE) == [(s, 1) for s in reversed(segments)] == oracle.segment_path(2 * p + 1)
    off, tokens = dev.segment_paths([5, 0, 6, 7, 3])                     # ids 6, 7: past the end -> no iterator, an empty row
    assert off.tolist() == [0, 5, 10, 10, 10, 15] and [(int(t) >> 1, int(t) & 1) for t in tokens[:5]] == [(7, 1), (6, 1), (3, 1), (2, 1), (0, 1)]
    starts = [1, 4, 5, 7, 8, 12]
    fwd = lambda a, b: [2 * v for v in range(a, b + 1)]
    rev = lambda a, b: [2 * v + 1 for v in range(b, a - 1, -1)]
    unit = fwd(1, 3) + fwd(4, 4) + rev(5, 6) + fwd(7, 7) + fwd(8, 11) + rev(12, 12)
    paths = [unit, fwd(2, 3) + fwd(4, 4), fwd(1, 2) + fwd(4, 4), fwd(1, 3) + rev(5, 5) + fwd(7, 7), fwd(4, 4) + fwd(9, 9), fwd(8, 9) + rev(9, 9), fwd(1, 3) + fwd(1, 3),
             unit * 800, [], fwd(12, 12), (fwd(1, 3) + rev(5, 6)) * 2000 + fwd(2, 2)]                # valid, broken in six ways, three chunks long, empty, long and broken at its very end
    dev, oracle = translated_graph(tmp_path, paths, starts, "segment_paths.gbz")
    ids = np.arange(2 * len(paths), dtype=np.uint64)
    off, tokens = dev.segment_paths(ids)
    for k, seq in enumerate(ids):
        want = oracle.segment_path(int(seq))
        got = [(int(t) >> 1, int(t) & 1) for t in tokens[int(off[k]):int(off[k + 1])]]
        assert got == want, (int(seq), got[:8], want[:8])
        assert dev.segment_path(int(seq) // 2, int(seq) & 1) == want
    assert len(oracle.segment_path(2 * 7)) == 6 * 800 and len(oracle.segment_path(2 * 1)) == 1                # (the long valid path; the path that enters a segment in its middle)
    plain = G.GBZ.load(os.path.join(O.GOLDEN, "example.gbz"))
    with pytest.raises(G.GbwtHipError) as e:
        plain.segment_paths([0])
    assert e.value.status == G._lib.BAD_ARGUMENT


def test_reference_samples_tag_in_the_header(tmp_path):
    """write_gfa_header (src/bin/gbunzip.rs:193-203): a GBWT with the `reference_samples` tag gets "H\tVN:Z:1.1\tRS:Z:<value>" -- in every
    path mode, the rest of the file as without the tag (this version of the reference reads the tag nowhere else on the path)."""
    s = S.Synth.chain(sites=300, haplotypes=24, alleles=2, model=S.MOSAIC, founders=4, switch_rate=0.02, seed=23)
    plain = tmp_path / "plain.gbz"
    s.save(str(plain), as_gbz=True)
    s.set_tag("reference_samples", "s0 s3")
    tagged = tmp_path / "tagged.gbz"
    s.save(str(tagged), as_gbz=True)
    dev, oracle, untagged = G.GBZ.load(str(tagged)), O.OracleGBZ(str(tagged)), G.GBZ.load(str(plain))
    for mode in (G.PATHS_DEFAULT, G.PATHS_PAN_SN, G.PATHS_REF_ONLY):
        out, ref = tmp_path / f"tagged{mode}.gfa", tmp_path / f"plain{mode}.gfa"
        dev.write_gfa(str(out), mode)
        untagged.write_gfa(str(ref), mode)
        got = out.read_bytes()
        assert got.startswith(b"H\tVN:Z:1.1\tRS:Z:s0 s3\nS\t"), got[:40]
        assert got == oracle.gfa(mode), mode
        assert got[len(b"H\tVN:Z:1.1\tRS:Z:s0 s3\n"):] == ref.read_bytes()[len(b"H\tVN:Z:1.1\n"):], mode
    ids = list(range(dev.paths()))
    for mode in (0, 1, 2):
        assert dev.path_lines(ids, mode) == oracle.path_lines(ids, mode) == untagged.path_lines(ids, mode)


def test_config_c4_shape_whole_file(tmp_path):
    """Config 4's shape: 24 contigs x 3 graph components each, ~2 300 ragged walks of 20 samples x 2 phases with non-zero
    fragment offsets, 24 generic paths -- the whole file in the three path modes and the W-lines in any order against the oracle
    (write_walks selects by sample, src/bin/gbunzip.rs:396-417; fragment and fragment + length, 508-519; PathName, src/gbwt.rs:912-970),
    and every sequence of the index, forward and reverse."""
    g = S.Synth.genome(contigs=24, fragments=3, haplotypes=40, sites=60, seed=11)
    assert g.paths > 2000
    path = tmp_path / "c4.gbz"
    g.save(str(path), as_gbz=True)
    dev, oracle = G.GBZ.load(str(path)), O.OracleGBZ(str(path))
    for mode in (G.PATHS_DEFAULT, G.PATHS_PAN_SN, G.PATHS_REF_ONLY):
        out = tmp_path / f"c4_{mode}.gfa"
        dev.write_gfa(str(out), mode)
        got, exp = out.read_bytes(), oracle.gfa(mode)
        assert hashlib.sha256(got).hexdigest() == hashlib.sha256(exp).hexdigest() and got == exp, mode
    text = oracle.gfa()
    walks = [l.split(b"\t") for l in text.split(b"\n") if l.startswith(b"W\t")]
    assert len({f[3] for f in walks}) == 24 and sum(int(f[4]) != 0 for f in walks) > 1000
    rng = np.random.default_rng(4)
    ids = rng.permutation(g.paths)[:700]
    for mode in (1, 2):
        assert dev.path_lines(ids, mode) == oracle.path_lines(ids, mode)
    seqs = np.arange(g.sequences, dtype=np.uint64)
    offsets, nodes = dev.sequences_csr(seqs)
    o_off, o_nodes = oracle.gbwt().extract(seqs, threads=8)
    assert np.array_equal(offsets, o_off) and np.array_equal(nodes, o_nodes)
    for p in (0, 1, g.paths // 2, g.paths - 1):
        assert np.array_equal(nodes[offsets[2 * p]:offsets[2 * p + 1]], g.path(p))


def test_whole_file_in_many_batches(tmp_path):
    """The pipelined writer (gbwt_hip_write_gfa_mode: byte-bounded batches, two device text buffers in turn, a writer thread with two
    pinned buffers) with a budget so small that every pass takes dozens of batches: the same bytes as the oracle's gbunzip restatement in
    the three path modes, and as the file written in one batch."""
    import subprocess
    import sys
    g = S.Synth.genome(contigs=6, fragments=4, haplotypes=24, sites=400, seed=5)
    path = tmp_path / "batches.gbz"
    g.save(str(path), as_gbz=True)
    oracle = O.OracleGBZ(str(path))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); import gbwt_rs_amd as G\n"
            f"dev = G.GBZ.load({str(path)!r})\n"
            f"for mode in (0, 1, 2): dev.write_gfa({str(tmp_path)!r} + '/small_' + str(mode) + '.gfa', mode)\n")
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, GBWT_HIP_GFA_BATCH_MIB="1"), timeout=600)
    dev = G.GBZ.load(str(path))
    for mode in (G.PATHS_DEFAULT, G.PATHS_PAN_SN, G.PATHS_REF_ONLY):
        exp = oracle.gfa(mode)
        assert len(exp) > (3 << 20) or mode == G.PATHS_REF_ONLY       # several 1 MiB batches
        assert (tmp_path / f"small_{mode}.gfa").read_bytes() == exp, mode
        out = tmp_path / f"one_{mode}.gfa"
        dev.write_gfa(str(out), mode)
        assert out.read_bytes() == exp, mode


def test_host_records_are_made_on_first_use(tmp_path):
    """An open from a file decodes the record starts on the device and leaves the HOST's image of the records (bytes + starts: what the S / L
    lines of the whole-file writer read, nothing an open or a lines request does) unmade (HostIndex::ensure_records; files of 4 MB or more with
    1 MB or more of record bytes -- smaller ones are decoded in the foreground as before).  The label lengths of the nodes that do not exist
    are zeroed on the device instead (k_mask_label_lengths).  Lines before and after, the whole file in the three modes, and the same from a
    process with GBWT_HIP_LAZY_HOST_RECORDS=0 -- all equal to the oracle's."""
    import subprocess
    import sys
    g = S.Synth.genome(contigs=8, fragments=4, haplotypes=48, sites=4000, seed=9)
    path = tmp_path / "lazy.gbz"
    g.save(str(path), as_gbz=True)
    assert os.path.getsize(path) >= (4 << 20) and len(g.data()) >= (1 << 20)
    oracle = O.OracleGBZ(str(path))
    ids = np.arange(g.paths, dtype=np.uint64)
    for flags in (G.OPEN_GFA, G.OPEN_ALL):
        dev = G.GBZ.load(str(path), flags=flags)
        before = dev.memory_usage()["index_host_bytes"]
        for mode in (0, 1, 2):
            assert dev.path_lines(ids, mode) == oracle.path_lines(ids, mode)
        assert dev.memory_usage()["index_host_bytes"] == before           # (a lines request reads none of it)
        for mode in (G.PATHS_DEFAULT, G.PATHS_PAN_SN, G.PATHS_REF_ONLY):
            out = tmp_path / f"lazy_{flags}_{mode}.gfa"
            dev.write_gfa(str(out), mode)
            assert out.read_bytes() == oracle.gfa(mode), (flags, mode)
        after = dev.memory_usage()["index_host_bytes"]
        assert after >= before + len(g.data()) + 8 * len(g.starts()), (before, after)    # made by the graph lines, once
        dev.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); import gbwt_rs_amd as G\n"
            f"dev = G.GBZ.load({str(path)!r})\n"
            f"assert dev.memory_usage()['index_host_bytes'] >= {len(g.data())}\n"
            f"dev.write_gfa({str(tmp_path)!r} + '/eager.gfa', 0)\n")
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, GBWT_HIP_LAZY_HOST_RECORDS="0"), timeout=600)
    assert (tmp_path / "eager.gfa").read_bytes() == oracle.gfa(0)


def test_config_c4_small_whole_file():
    """Config 4's stand-in of rounds 3-4 (bench.py's `config4_small`): Synth.genome 24 contigs x 20 components, 90 haplotypes -- 32 286
    ragged walks over one-base nodes, 0.50 G LF-steps, 4.5 GB of GFA.  The whole file through the pipelined writer (1 GiB batches, 64 MiB pieces), checked
    piece by piece:
      * H-, S-, L- and P-lines: byte-identical to the oracle's gbunzip restatement in ref-only mode (src/bin/gbunzip.rs:205-332);
      * the W-lines: identical to the text of ONE device request for all walks; a seeded sample of 2 048 of them byte-identical to the
        oracle's path_to_w_line (src/bin/gbunzip.rs:495-550); EVERY line's header fields and length against the generator's ground truth
        (sample, phase, contig, fragment, fragment + length, one token per node);
      * the extraction behind it: per-path checksums of all 32 286 forward sequences against the generator."""
    g = S.Synth.genome(contigs=24, fragments=20, haplotypes=90, sites=6000, seed=42)
    assert g.paths > 30000
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path, out = os.path.join(tmp, "gbwt_c4_full.gbz"), os.path.join(tmp, "gbwt_c4_full.gfa")
    try:
        g.save(path, as_gbz=True)
        dev, oracle = G.GBZ.load(path), O.OracleGBZ(path)
        generic = np.array(g.generic_paths(), dtype=np.uint64)
        walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
        # the extraction: every forward sequence against the generator
        ids = 2 * np.arange(g.paths, dtype=np.uint64)
        res = dev.extract_device(ids)
        assert int(res.total) == (g.size - g.sequences) // 2
        truth = np.array([g.path_checksum(p) for p in range(g.paths)], dtype=np.uint64)
        assert np.array_equal(dev.path_sums(len(ids)), truth)
        # the whole file
        dev.write_gfa(out)
        head = oracle.gfa(G.PATHS_REF_ONLY)
        whole = np.memmap(out, dtype=np.uint8, mode="r")
        assert len(whole) > len(head) and bytes(whole[:len(head)]) == head, "H/S/L/P lines differ from the oracle"
        text = dev.path_lines(walks, 1)                        # one request for all walks
        assert len(whole) == len(head) + len(text)
        tail = np.frombuffer(text, dtype=np.uint8)
        step = 1 << 28
        for lo in range(0, len(tail), step):
            assert np.array_equal(whole[len(head) + lo:len(head) + lo + step], tail[lo:lo + step]), "the batched file differs from the single request"
        ends = np.concatenate([np.flatnonzero(tail[lo:lo + step] == 10) + lo for lo in range(0, len(tail), step)])
        assert len(ends) == len(walks)
        starts = np.concatenate([[0], ends[:-1] + 1])
        # every line: header fields and length from the generator's ground truth
        samples = [f"s{k}" for k in range((90 + 1) // 2)] + ["_gbwt_ref"]
        pow10 = 10 ** np.arange(1, 10, dtype=np.uint64)
        for k, p in enumerate(walks):
            nodes = g.path(int(p))
            sample, contig, phase, fragment = (int(x) for x in g.path_names[int(p)])
            header = f"W\t{samples[sample]}\t{phase}\tchr{contig + 1}\t{fragment}\t{fragment + len(nodes)}\t".encode()
            lo, hi = int(starts[k]), int(ends[k])
            assert text[lo:lo + len(header)] == header, (k, int(p))
            digits = 1 + np.searchsorted(pow10, (nodes >> 1).astype(np.uint64), side="right")
            assert hi - lo == len(header) + int(digits.sum()) + len(nodes), (k, int(p))
        # a seeded sample of lines against the oracle, byte for byte
        rng = np.random.default_rng(2024)
        for k in np.sort(rng.choice(len(walks), 2048, replace=False)):
            assert text[int(starts[k]):int(ends[k]) + 1] == oracle.path_lines([int(walks[k])], 1), int(walks[k])
        del whole
    finally:
        for f in (path, out):
            if os.path.exists(f):
                os.remove(f)


def test_config_c4_full_size():
    """BASELINE config 4 at the size SURVEY 8(d) states (tools/c4_bench.py: SIZES["full"], bench.py's `config4`): 24 contigs x 20 graph
    components walked by 90 haplotypes = ~42 000 ragged walks over ~109 M node ids with labels of 1 .. 1 024 bp, 5.7 G LF-steps, 51 GB of
    W-lines -- the walks of the last contig start just below 2^32, so their end coordinates (fragment + summed label lengths,
    src/bin/gbunzip.rs:532-540) need more than 32 bits.
      * the extraction: per-path checksums of ALL forward sequences against the generator, and length / sum / order-dependent hash of
        ALL of them against the oracle's walk (src/gbwt.rs:557-568);
      * ONE device request for all W-lines: EVERY line's length and header fields (sample, phase, contig, fragment, end) against the
        generator's ground truth (path_text_stats, itself checked against the oracle in tests/test_dist_cpu.py);
      * a seeded sample of 192 W-lines and all P-lines byte for byte against the oracle (path_to_w_line / write_p_line,
        src/bin/gbunzip.rs:438-550);
      * the whole file through the pipelined writer to /dev/shm: its size, and its P/W part against the device text at 96 seeded places
        (H/S/L lines are compared with the oracle at the small size: test_config_c4_small_whole_file -- the same host code)."""
    import sys
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from gbwt_rs_amd import dist as D
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import c4_bench
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    path, out = os.path.join(tmp, "gbwt_c4_stated.gbz"), os.path.join(tmp, "gbwt_c4_stated.gfa")
    cores = min(os.cpu_count() or 1, 64)
    try:
        g = c4_bench.generate("full", path)
        assert g.paths > 40000 and g.alphabet_size // 2 > 85000000
        dev, oracle = G.GBZ.load(path), O.OracleGBZ(path)
        generic = np.array(g.generic_paths(), dtype=np.uint64)
        walks = np.setdiff1d(np.arange(g.paths, dtype=np.uint64), generic)
        with ThreadPoolExecutor(cores) as pool:
            truth = np.array(list(pool.map(g.path_checksum, range(g.paths))), dtype=np.uint64)
            stats = np.array(list(pool.map(g.path_text_stats, range(g.paths))), dtype=np.int64)      # nodes, digits, bp per path
        # the extraction
        ids = 2 * np.arange(g.paths, dtype=np.uint64)
        res = dev.extract_device(ids)
        assert int(res.total) == (g.size - g.sequences) // 2 == int(stats[:, 0].sum())
        assert np.array_equal(dev.path_sums(len(ids)), truth)
        hashes, lens = dev.path_hashes(len(ids)), np.diff(dev.last_offsets(len(ids)))
        assert np.array_equal(lens, stats[:, 0].astype(np.uint64))
        # (every path, not a sample -- the oracle walks all 42 016 forward sequences, 5.7 G LF-steps, about a minute on 64 threads)
        o_steps, o_lens, o_sums, o_hashes = oracle.gbwt().extract_checksums(ids, cores)
        assert o_steps == int(res.total)
        assert np.array_equal(hashes, o_hashes) and np.array_equal(lens, o_lens) and np.array_equal(truth, o_sums), "a row differs from the oracle's walk"
        # one request for all W-lines: every line's length and header against the generator
        lines = dev.path_lines_device(walks, 1)
        device = torch.device("cuda", 0)
        off, text = D.lines_tensors(lines, device)
        assert int(lines.total) > 45 << 30 and off.numel() == len(walks) + 1
        names = np.array([g.path_names[int(p)] for p in walks], dtype=np.int64)                      # sample, contig, phase, fragment
        headers = [f"W\t{g.sample_names[s]}\t{ph}\tchr{c + 1}\t{f}\t{f + bp}\t".encode() for (s, c, ph, f), bp in zip(names, stats[walks.astype(np.int64), 2])]
        assert sum(f + bp > 1 << 32 for (_, _, _, f), bp in zip(names, stats[walks.astype(np.int64), 2])) >= 64, "no end coordinate past 2^32"
        want_len = np.array([len(h) for h in headers], dtype=np.int64) + stats[walks.astype(np.int64), 1] + stats[walks.astype(np.int64), 0] + 1
        assert np.array_equal(np.diff(off.cpu().numpy()), want_len), "line lengths differ from the generator's ground truth"
        heads = text[(off[:-1, None] + torch.arange(96, device=device)[None, :]).clamp_(max=text.numel() - 1)].cpu().numpy()
        for k, h in enumerate(headers):
            assert heads[k, :len(h)].tobytes() == h, (k, int(walks[k]))
        host_off = off.cpu().numpy()
        assert int(text[int(host_off[-1]) - 1]) == 10 and bool((text[(off[1:] - 1)] == 10).all())           # every line ends with a newline
        # a seeded sample of W-lines and all P-lines against the oracle, byte for byte
        for k in np.sort(np.random.default_rng(2025).choice(len(walks), 192, replace=False)):
            got = text[int(host_off[k]):int(host_off[k + 1])].cpu().numpy().tobytes()
            assert got == oracle.path_lines([int(walks[k])], 1), int(walks[k])
        assert dev.another_workspace().path_lines(generic, 0) == oracle.path_lines([int(p) for p in generic], 0)
        # the whole file (50 batches of 1 GiB through the writer), its P/W part against the single request at seeded places
        p_text_len = len(dev.another_workspace().path_lines(generic, 0))
        keep = text.clone()                                                                          # (the writer uses this workspace's text buffers)
        dev.write_gfa(out)
        size = os.path.getsize(out)
        head_len = size - p_text_len - int(keep.numel())
        assert head_len > 0
        whole = np.memmap(out, dtype=np.uint8, mode="r")
        assert bytes(whole[:2]) == b"H\t" and whole[head_len - 1] == 10 and bytes(whole[head_len:head_len + 2]) == b"P\t"
        w_at = head_len + p_text_len
        assert bytes(whole[w_at:w_at + 2]) == b"W\t" and whole[-1] == 10
        gen = np.random.default_rng(7)
        for lo in list(gen.integers(0, keep.numel() - (1 << 20), 94)) + [0, keep.numel() - (1 << 20)]:
            assert np.array_equal(whole[w_at + int(lo):w_at + int(lo) + (1 << 20)], keep[int(lo):int(lo) + (1 << 20)].cpu().numpy()), int(lo)
        del whole, keep
        dev.close()
    finally:
        for f in (path, path + ".generic.npy", out):
            if os.path.exists(f):
                os.remove(f)


@pytest.mark.parametrize("fill_mode", [0, 1, 2])
def test_line_cache_serves_every_mode(tmp_path, monkeypatch, fill_mode):
    """The line cache of the index: the sizes of every path's line -- token bytes chunk by chunk, summed label lengths -- are found by ONE
    walk at open (round 6: a walker per segment between two sequence samples; round 5 left them behind the first request of a path), and
    every request sizes and places its lines from them without a sizing pass, in ANY line mode (a P-line's text is the W-line's plus
    separators).  Lines of paths longer than one chunk (chunk boundaries inside and at the end of a segment), paths that visit nodes
    again, empty paths, duplicates and subsets in another order: asked for in all three modes, whichever comes first, against the oracle
    and against a handle with the cache switched off (GBWT_HIP_LINE_CACHE=0: every request sizes its lines itself)."""
    paths = [[2 * (1 + (7 * k + j) % 50) + ((k + j) % 3 == 0) for j in range(ln)] for k, ln in enumerate([0, 1, 9000, 4096, 4097, 12289, 5, 0, 8192, 300])]
    s = S.Synth.from_paths(paths, bidirectional=True).attach_gbz(seed=3)
    path = str(tmp_path / "cache.gbz")
    s.save(path, as_gbz=True)
    oracle = O.OracleGBZ(path)
    dev = G.GBZ.load(path)
    monkeypatch.setenv("GBWT_HIP_LINE_CACHE", "0")
    plain = G.GBZ.load(path)
    monkeypatch.delenv("GBWT_HIP_LINE_CACHE")
    everything = list(range(len(paths)))
    first = [2, 0, 5, 9]
    assert dev.open_times()["line_sizes_ms"] > 0 and plain.open_times()["line_sizes_ms"] == 0
    assert dev.path_lines(first, fill_mode) == oracle.path_lines(first, fill_mode)
    for mode in (0, 1, 2):
        for ids in (first, [5, 5, 2], everything, everything[::-1], [7], [3, 2]):
            got = dev.path_lines(ids, mode)
            assert got == oracle.path_lines(ids, mode) == plain.path_lines(ids, mode), (fill_mode, mode, ids)
    out = tmp_path / "whole.gfa"
    dev.write_gfa(str(out))
    assert out.read_bytes() == oracle.gfa()


def test_line_cache_filled_by_concurrent_requests(tmp_path):
    """Several host threads, each with a workspace of its own, format overlapping sets of paths of ONE handle at the same time, in different
    line modes, all of them sizing their lines from the handle's line cache (read-only after the open): every text equals the oracle's."""
    import threading
    s = S.Synth.chain(sites=2500, haplotypes=160, alleles=2, model=S.MOSAIC, founders=8, switch_rate=0.01, seed=19, extra=1, indel_every=5)
    path = str(tmp_path / "threads.gbz")
    s.save(path, as_gbz=True)
    oracle = O.OracleGBZ(path)
    dev = G.GBZ.load(path, flags=G.OPEN_GFA)
    everything = list(range(s.paths))
    jobs = [(everything, 1), (everything[::-1], 0), (everything[::2], 2), (everything[40:120], 1), ([3, 3, 77], 0), (everything, 2)]
    want = [oracle.path_lines(ids, mode) for ids, mode in jobs]
    failures = []

    def work(k):
        try:
            view = dev.another_workspace()
            for _ in range(3):
                ids, mode = jobs[k]
                if view.path_lines(ids, mode) != want[k]:
                    failures.append(k)
            view.close()
        except Exception as e:      # noqa: BLE001
            failures.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures


def test_bare_gbwt_has_no_gfa():
    dev = G.GBZ.load(os.path.join(O.GOLDEN, "example.gbwt"))
    with pytest.raises(G.GbwtHipError):
        dev.path_lines([0], 1)


@pytest.mark.parametrize("alleles,sites,haps", [(2, 700, 300), (5, 90, 120), (2, 4500, 12)])   # the last one: 9 000 nodes per line = three chunks
def test_synthetic_gfa_matches_oracle(tmp_path, alleles, sites, haps):
    """Whole-file parity on generated GBZ files: node ids with 1-4 digits, single P-line, many W-lines, chunked lines
    (paths longer than one formatting chunk)."""
    s = S.Synth.chain(sites=sites, haplotypes=haps, alleles=alleles, model=S.MOSAIC, founders=8, switch_rate=0.02, seed=17)
    path = tmp_path / "synth.gbz"
    s.save(str(path), as_gbz=True)
    dev, oracle = G.GBZ.load(str(path)), O.OracleGBZ(str(path))
    out = tmp_path / "synth.gfa"
    dev.write_gfa(str(out))
    got, exp = out.read_bytes(), oracle.gfa()
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(exp).hexdigest()
    assert got == exp
    ids = [haps - 1, 1, 7]
    assert dev.path_lines(ids, 1) == oracle.path_lines(ids, 1)
    assert dev.path_lines([0], 0) == oracle.path_lines([0], 0)


def test_tokens_on_the_device(tmp_path):
    """csrc/gfa_tokens.hpp on the GPU itself (the byte-align instruction, the 24-bit multiplications): tests/cpp/token_device_check.hip makes the
    tokens of a million ids up to 2^31 - 1 -- ten digits: no index small enough for a test has such ids -- in every form on the device and
    compares them with snprintf."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "token_device_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(root, "gbwt_rs_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "token_device_check.hip"), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "device tokens checked" in out.stdout


@pytest.mark.parametrize("chop", [1, 3])
def test_token_text_at_chunk_and_digit_boundaries(tmp_path, chop):
    """The formatter's whole-token form (tokens made in registers, OR-ed into a zeroed staging buffer) against the oracle: P- and W-lines of paths
    of several chunks, single paths, the first / last positions of chunks, node ids of 1-5 digits, from two workspaces of one handle."""
    s = S.Synth.chain(sites=2600, haplotypes=40, alleles=3, model=S.MOSAIC, founders=6, switch_rate=0.01, seed=23, chop=chop)
    path = tmp_path / "forms.gbz"
    s.save(str(path), as_gbz=True)
    dev, oracle = G.GBZ.load(str(path)), O.OracleGBZ(str(path))
    paths = list(range(dev.paths()))
    texts = []
    for _ in range(2):
        ws = dev.another_workspace()
        texts.append((ws.path_lines(paths, 1), ws.path_lines(paths, 0), ws.path_lines(paths[3:4], 1), ws.path_lines(paths[:1], 0)))
        ws.close()
    assert texts[0] == texts[1]
    assert texts[0][0] == oracle.path_lines(paths, 1) and texts[0][1] == oracle.path_lines(paths, 0)
    assert texts[0][2] == oracle.path_lines(paths[3:4], 1) and texts[0][3] == oracle.path_lines(paths[:1], 0)
    dev.close()


def test_device_resident_lines_and_rccl_gather(tmp_path):
    """gbwt_hip_path_lines_device leaves the text in HBM; dist.lines_tensors wraps it without a copy and
    dist.gather_lines moves it over RCCL (backend "nccl", world size 1 on this box: the collective and the tensor
    plumbing are the real ones, the peers are missing).  Runs in a child process: a process group is per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
import gbwt_rs_amd as G
from gbwt_rs_amd import dist as D
import oracle_lib as O
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=PORT, RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
path = os.path.join(O.GOLDEN, "example.gbz")
dev, oracle = G.GBZ.load(path), O.OracleGBZ(path)
ids = [5, 2, 3]
want = oracle.path_lines(ids, 1)
lines = dev.path_lines_device(ids, 1)
off, text = D.lines_tensors(lines, torch.device("cuda", 0))
assert text.is_cuda and bytes(text.cpu().numpy().tobytes()) == want and int(off[-1]) == len(want) and off.numel() == 4
assert dev.path_lines(ids, 1) == want                     # the copy-out of the same request
for interleaved in (False, True):
    g_off, g_text = D.gather_lines(off, text, dst=0, interleaved=interleaved)
    assert bytes(g_text.cpu().numpy().tobytes()) == want and int(g_off[-1]) == len(want)
# node ids the same way
p = dev.extract_device(np.array([0, 2, 4], dtype=np.uint64))
o, n = D.paths_tensors(p, torch.device("cuda", 0))
oo, nn = oracle.gbwt().extract(np.array([0, 2, 4], dtype=np.uint64))
assert np.array_equal(o.cpu().numpy(), oo.astype(np.int64)) and np.array_equal(n.cpu().numpy().astype(np.uint32), nn)
g_off, g_val = D.gather_rows(o[1:] - o[:-1], n, dst=0)
assert np.array_equal(g_val.cpu().numpy().astype(np.uint32), nn)
dist.destroy_process_group()
print("RCCL_GATHER_OK")
'''
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}; PORT = '{port}'\n" + code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "RCCL_GATHER_OK" in out.stdout, out.stderr[-3000:]
