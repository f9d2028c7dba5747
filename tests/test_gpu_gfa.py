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


def test_translation_graph_is_rejected_not_wrong():
    dev = G.GBZ.load(os.path.join(O.GOLDEN, "translation.gbz"))
    with pytest.raises(G.GbwtHipError) as e:
        dev.path_lines([0], 1)
    assert e.value.status == 7   # GBWT_HIP_UNSUPPORTED


def test_bare_gbwt_has_no_gfa():
    dev = G.GBZ.load(os.path.join(O.GOLDEN, "example.gbwt"))
    with pytest.raises(G.GbwtHipError):
        dev.path_lines([0], 1)


@pytest.mark.parametrize("alleles,sites,haps", [(2, 700, 300), (5, 90, 120)])
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
