"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/gbwt_hip.h declares,
parses/validates files on the host, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import gbwt_rs_amd as G
from gbwt_rs_amd import _lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-C", _lib.CSRC], stdout=subprocess.DEVNULL)


def declared_symbols():
    text = open(_lib.HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gbwt_hip_[a-z_]+)\s*\(", text)))


def test_header_symbols_exported():
    names = declared_symbols()
    assert len(names) >= 20
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/gbwt_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signatures out of sync with the header"


def test_parse_fixtures():
    st = G.parse_file(os.path.join(GOLDEN, "example.gbwt"))
    assert (st.size, st.sequences, st.alphabet_size, st.alphabet_offset, st.records, st.data_bytes) == (68, 12, 52, 21, 31, 141)
    assert st.bidirectional and st.has_metadata and not st.is_gbz and st.paths == 6
    st = G.parse_file(os.path.join(GOLDEN, "with-empty.gbwt"))
    assert (st.size, st.sequences, st.has_metadata) == (70, 14, 0)
    for name in ("example.gbz", "example-v1.gbz"):
        st = G.parse_file(os.path.join(GOLDEN, name))
        assert st.is_gbz and not st.has_translation and st.records == 31
    for name in ("translation.gbz", "translation-v1.gbz"):
        st = G.parse_file(os.path.join(GOLDEN, name))
        assert st.is_gbz and st.has_translation and (st.size, st.sequences, st.alphabet_size, st.alphabet_offset) == (48, 6, 24, 1)


def test_parse_rejects_what_the_reference_rejects(tmp_path):
    raw = bytearray(open(os.path.join(GOLDEN, "example.gbwt"), "rb").read())
    cases = {
        "tag": (0, raw[0] ^ 0xFF, "Invalid tag"),               # src/headers.rs:102-104
        "version": (4, 4, "Invalid version"),                    # src/headers.rs:105-114
        "flags": (40, 0x0F, "Invalid flags"),
        "sdsl": (40, 0x03, "SDSL"),                              # src/headers.rs:229-231
        "mismatch": (42 * 8, 140, "mismatch"),                   # src/bwt.rs:179-181
    }
    for name, (pos, value, msg) in cases.items():
        bad = bytearray(raw)
        bad[pos] = value
        p = tmp_path / f"{name}.gbwt"
        p.write_bytes(bad)
        with pytest.raises(G.GbwtHipError) as e:
            G.parse_file(str(p))
        assert e.value.status == _lib.INVALID_DATA and msg in str(e.value)
    p = tmp_path / "truncated.gbwt"
    p.write_bytes(raw[:800])
    with pytest.raises(G.GbwtHipError) as e:
        G.parse_file(str(p))
    assert e.value.status == _lib.INVALID_DATA
    with pytest.raises(G.GbwtHipError) as e:
        G.parse_file(str(tmp_path / "missing.gbwt"))
    assert e.value.status == _lib.IO_ERROR
    # a GBZ whose GBWT is not bidirectional (src/gbz.rs:684-686)
    z = bytearray(open(os.path.join(GOLDEN, "example-v1.gbz"), "rb").read())
    z[30 * 8] = 0x06
    p = tmp_path / "unidirectional.gbz"
    p.write_bytes(z)
    with pytest.raises(G.GbwtHipError) as e:
        G.parse_file(str(p))
    # GBWT::load already trips over the path count (src/gbwt.rs:424-428) before GBZ::load can object
    assert e.value.status == _lib.INVALID_DATA and ("not bidirectional" in str(e.value) or "path count" in str(e.value))


def test_no_cpu_fallback():
    """Without a GPU the product path fails loudly instead of computing on the host."""
    if G.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(G.GbwtHipError) as e:
        G.GBWT.load(os.path.join(GOLDEN, "example.gbwt"))
    assert e.value.status == _lib.NO_DEVICE


def test_product_does_not_link_the_oracle():
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.dirname(_lib.CSRC)):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "liboracle" not in src and "oracle_lib" not in src and "gbwt_oracle" not in src, f


def test_cpp_mirror_compiles_and_reports_no_device(tmp_path):
    """include/gbwt_hip.hpp (the C++ mirror of the reference's GBWT / GBZ interface) compiles with a plain host compiler
    against the C ABI; without a GPU the test program must see GBWT_HIP_NO_DEVICE (no CPU fallback)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "test_reference_api"
    csrc = os.path.join(root, "gbwt_rs_amd", "csrc")
    subprocess.run([shutil.which("g++") or "g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-o", str(exe),
                    os.path.join(root, "tests", "cpp", "test_reference_api.cpp"), "-L", csrc, "-lgbwt_hip", "-Wl,-rpath," + csrc], check=True)
    out = subprocess.run([str(exe), os.path.join(root, "tests", "golden")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    import gbwt_rs_amd as G
    if G.device_count() == 0:
        assert "GBWT_HIP_NO_DEVICE" in out.stdout
