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


MUTATIONS = (0xFFFFFFFFFFFFFFFF, 0x8000000000000000, 0xFFFFFFFFFFFFFFC1, 1 << 40, 0x7FFFFFFFFFFFFFFF, 0, 1, 65)


def mutated_files(tmp_path, names=None, values=MUTATIONS):
    """Every 64-bit element of every golden file overwritten with each of `values` (length words near 2^64 wrap
    (len + 7) / 8 and len * width; small ones truncate or shift every structure behind them)."""
    import glob
    target = str(tmp_path / "mutated.bin")
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.gb*"))):
        if names is not None and os.path.basename(path) not in names:
            continue
        raw = open(path, "rb").read()
        for value in values:
            for w in range(len(raw) // 8):
                if raw[8 * w:8 * w + 8] == value.to_bytes(8, "little"):
                    continue
                bad = bytearray(raw)
                bad[8 * w:8 * w + 8] = value.to_bytes(8, "little")
                with open(target, "wb") as f:
                    f.write(bad)
                yield os.path.basename(path), w, value, target


def mutated_large_file(tmp_path, per_region=40, values=(0xFFFFFFFFFFFFFFFF, 0, 1 << 40)):
    """The same for a generated GBZ of several MiB -- large enough for the threaded paths of an open (the early host-to-device copy of
    the record bytes out of the mapped file, the deferred decodes on threads, the background copy + label decode behind finish()):
    seeded word positions in the head of the file (headers, tags, the Elias-Fano index of the record starts), in its tail (node
    labels, translation) and in between (record bytes, metadata)."""
    import numpy as np
    from gbwt_rs_amd import synth as S
    s = S.Synth.chain(sites=40000, haplotypes=600, alleles=2, model=S.MOSAIC, founders=16, switch_rate=0.01, seed=77)
    source, target = str(tmp_path / "large.gbz"), str(tmp_path / "large_mutated.gbz")
    s.save(source, as_gbz=True)
    raw = open(source, "rb").read()
    words = len(raw) // 8
    assert len(raw) > (2 << 20)
    rng = np.random.default_rng(99)
    picks = np.concatenate([rng.integers(0, words // 12, per_region), rng.integers(words - words // 10, words, per_region),
                            rng.integers(words // 12, words - words // 10, per_region // 2), np.arange(0, 24)])
    for w in sorted(set(int(x) for x in picks)):
        for value in values:
            if raw[8 * w:8 * w + 8] == value.to_bytes(8, "little"):
                continue
            bad = bytearray(raw)
            bad[8 * w:8 * w + 8] = value.to_bytes(8, "little")
            with open(target, "wb") as f:
                f.write(bad)
            yield w, value, target


def test_corrupt_words_of_a_large_file_are_invalid_data(tmp_path):
    accepted = rejected = 0
    reasons = set()
    for w, value, path in mutated_large_file(tmp_path):
        try:
            G.parse_file(path)
            accepted += 1
        except G.GbwtHipError as e:
            assert e.status == _lib.INVALID_DATA, (w, hex(value), str(e))
            rejected += 1
            reasons.add(str(e).split(":")[1].strip()[:40])
    assert rejected > 40 and accepted > 20 and len(reasons) >= 4, (accepted, rejected, reasons)


def test_corrupt_words_are_invalid_data_never_an_abort(tmp_path):
    """A corrupt length word must come back as GBWT_HIP_INVALID_DATA (io::ErrorKind::InvalidData in the reference's
    loaders), not as an exception crossing the C ABI (std::terminate) or an out-of-bounds read.  Runs in-process: an
    abort fails the whole suite, which is the point."""
    accepted = rejected = 0
    for name, w, value, path in mutated_files(tmp_path):
        try:
            G.parse_file(path)
            accepted += 1
        except G.GbwtHipError as e:
            assert e.status == _lib.INVALID_DATA, (name, w, hex(value), str(e))
            rejected += 1
    assert rejected > 5000 and accepted > 1000, (accepted, rejected)   # payload words (names, record bytes) are free to change


def test_zstd_labels_are_bounded_by_the_stream(tmp_path):
    """The declared length of the zstd-compressed node labels (graph version 4) is checked against what the stream
    produces, whatever the word says -- it never sizes an allocation on its own."""
    raw = open(os.path.join(GOLDEN, "example.gbz"), "rb").read()
    st = G.parse_file(os.path.join(GOLDEN, "example.gbz"))
    assert st.is_gbz
    hits = 0
    for w in range(len(raw) // 8):
        for value in (1 << 62, 1 << 36):
            bad = bytearray(raw)
            bad[8 * w:8 * w + 8] = value.to_bytes(8, "little")
            p = tmp_path / "z.gbz"
            p.write_bytes(bad)
            try:
                G.parse_file(str(p))
            except G.GbwtHipError as e:
                assert e.status == _lib.INVALID_DATA
                hits += "Decompressed string length" in str(e)
    assert hits >= 2


def test_loader_under_sanitizers(tmp_path):
    """The same mutations against the loader built with -fsanitize=address,undefined (host compiler, CPU only)."""
    import glob
    import shutil
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "mutate_loader"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([gxx, "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", _lib.CSRC,
                    os.path.join(root, "tests", "cpp", "mutate_loader.cpp"), os.path.join(_lib.CSRC, "host_index.cpp"), "-ldl", "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe), str(tmp_path / "scratch.bin")] + sorted(glob.glob(os.path.join(GOLDEN, "*.gb*"))), capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "accepted" in out.stdout


def test_walk_kernel_keeps_four_waves_per_simd():
    """k_walk_direct is sized for four waves per SIMD = eight workgroups per CU: 128 VGPRs at most and nothing spilled.
    Round 4 lost a tenth of the headline for a few commits to a table decoder that kept sixteen more registers alive (135 VGPRs, three
    waves per SIMD) without any test noticing; hipcc's resource remarks need no GPU."""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull,
                          os.path.join(_lib.CSRC, "walk_direct.hip")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stderr.splitlines()
    at = next(i for i, l in enumerate(lines) if "Function Name" in l and "k_walk_direct" in l)
    block = "\n".join(lines[at:at + 14])
    field = lambda name: int(re.search(name + r": (\d+)", block).group(1))
    assert field("    VGPRs") <= 128 and field("VGPRs Spill") == 0 and field(r"Occupancy \[waves/SIMD\]") >= 4, block


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


def test_product_library_holds_no_test_transport():
    """The loopback transport (ranks as threads of one process) and the self-send switch are test infrastructure: compiled only into
    libgbwt_hip_testtransport.so (-DGBWT_HIP_TEST_TRANSPORT), never into the product library -- a stray environment variable cannot
    replace RCCL with same-device copies."""
    csrc = _lib.CSRC
    product = open(os.path.join(csrc, "libgbwt_hip.so"), "rb").read()
    for needle in (b"GBWT_HIP_COMM_LOOPBACK", b"GBWT_HIP_COMM_SELF_SEND", b"loopback"):
        assert needle not in product, needle
    test_build = open(os.path.join(csrc, "libgbwt_hip_testtransport.so"), "rb").read()
    assert b"GBWT_HIP_COMM_LOOPBACK" in test_build and b"GBWT_HIP_COMM_SELF_SEND" in test_build


def test_gfa_tokens_against_snprintf(tmp_path):
    """The register arithmetic that makes GFA node tokens (csrc/gfa_tokens.hpp: digits by multiplications, the token OR-ed in as aligned dwords)
    compiled for the host under ASan + UBSan: every token length, both orientations, W-line / P-line / first-of-P-line forms, both builders,
    every alignment, against snprintf (the reference prints ids with Rust's Display, src/bin/gbunzip.rs:462-476, 542-547)."""
    import shutil
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "token_check"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([gxx, "-std=c++17", "-O2", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", _lib.CSRC,
                    os.path.join(root, "tests", "cpp", "token_check.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert int(re.search(r"tokens checked: (\d+)", out.stdout).group(1)) > 1000000


def test_format_kernel_keeps_eight_waves_per_simd():
    """k_format_chunks hides its load -> scan -> stage -> store chain behind eight workgroups per CU (eight positions per thread = five waves
    per SIMD measured 23 % slower, profiles/r05_format_stream.txt): at most 64 VGPRs, nothing spilled, LDS for eight workgroups."""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull,
                          os.path.join(_lib.CSRC, "gfa.hip")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stderr.splitlines()
    at = next(i for i, l in enumerate(lines) if "Function Name" in l and "k_format_chunksILj4E" in l)
    block = "\n".join(lines[at:at + 14])
    field = lambda name: int(re.search(name + r": (\d+)", block).group(1))
    assert field("    VGPRs") <= 64 and field("VGPRs Spill") == 0 and field(r"Occupancy \[waves/SIMD\]") == 8 and field(r"LDS Size \[bytes/block\]") <= 160 * 1024 // 8, block
