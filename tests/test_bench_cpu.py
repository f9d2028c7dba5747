"""Host logic of bench.py that needs no GPU: the fingerprint that ties a PMC profile to a build, and the rank launcher."""
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_fingerprint_follows_sources_and_knobs(monkeypatch):
    for k in list(os.environ):
        if k.startswith("GBWT_HIP_"):
            monkeypatch.delenv(k)
    a = bench.source_fingerprint()
    assert a == bench.source_fingerprint() and len(a) == 16
    monkeypatch.setenv("GBWT_HIP_RING_SLOTS", "32")
    b = bench.source_fingerprint()
    assert b != a                                   # another knob setting is another measurement
    monkeypatch.setenv("GBWT_HIP_LIB", "/somewhere/else.so")
    assert bench.source_fingerprint() == b          # where the library is loaded from is not a knob of the kernel


def test_committed_traffic_profile_names_its_build_and_workload():
    """profiles/*_hbm_traffic.json of this round carry the fingerprint and the workload bench.py matches them by."""
    path = os.path.join(ROOT, "profiles", "r03_hbm_traffic.json")
    t = json.load(open(path))
    assert len(t["source_fingerprint"]) == 16 and t["workload_key"] == "sites=333334 haplotypes=5000 model=mosaic seed=42"
    assert t["traffic_bytes_per_launch"] == 2 * t["fetch_bytes_raw"] + t["write_bytes"]
    assert 13.3e9 < t["write_bytes"] < 13.5e9       # every node id once


def test_round4_profiles_cover_every_config():
    """Round 4: every BASELINE config has a PMC traffic file that bench.py can match (fingerprint + workload key) and the per-kernel split."""
    keys = {"r04_hbm_traffic.json": "sites=333334 haplotypes=5000 model=mosaic seed=42", "r04_secondary_hbm_traffic.json": "secondary",
            "r04_high_degree_hbm_traffic.json": "high_degree", "r04_search_hbm_traffic.json": "search", "r04_config4_hbm_traffic.json": "config4"}
    prints = set()
    for name, key in keys.items():
        t = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert t["workload_key"] == key and len(t["source_fingerprint"]) == 16, name
        assert t["traffic_bytes_per_launch"] == 2 * t["fetch_bytes_raw"] + t["write_bytes"] and t["kernels"], name
        assert abs(sum(k["traffic_bytes_per_launch"] for k in t["kernels"].values()) - t["traffic_bytes_per_launch"]) < 1.0, name
        prints.add(t["source_fingerprint"])
    assert len(prints) == 1                         # one build, one gpurun call
    assert set(json.load(open(os.path.join(ROOT, "profiles", "r04_config4_hbm_traffic.json")))["kernels"]) == {"k_walk_direct", "k_chunk_stats", "k_format_chunks"}


def test_round5_profiles_cover_every_config():
    """Round 5: every BASELINE config (config 4 at its stated size AND as the small stand-in) has a PMC traffic file of ONE build; the search
    profile takes FETCH_SIZE as it is (scattered 64-byte requests are counted exactly: profiles/r05_fetch_calibration.txt), the others x 2."""
    keys = {"r05_hbm_traffic.json": "sites=333334 haplotypes=5000 model=mosaic seed=42", "r05_secondary_hbm_traffic.json": "secondary",
            "r05_high_degree_hbm_traffic.json": "high_degree", "r05_search_hbm_traffic.json": "search", "r05_config4_hbm_traffic.json": "config4",
            "r05_config4_small_hbm_traffic.json": "config4_small"}
    prints = set()
    for name, key in keys.items():
        t = json.load(open(os.path.join(ROOT, "profiles", name)))
        factor = t.get("fetch_factor", 2.0)
        assert t["workload_key"] == key and len(t["source_fingerprint"]) == 16, name
        assert factor == (1.0 if key == "search" else 2.0), name
        assert t["traffic_bytes_per_launch"] == factor * t["fetch_bytes_raw"] + t["write_bytes"] and t["kernels"], name
        prints.add(t["source_fingerprint"])
    assert len(prints) == 1
    c4 = json.load(open(os.path.join(ROOT, "profiles", "r05_config4_hbm_traffic.json")))
    assert set(c4["kernels"]) == {"k_walk_direct", "k_format_chunks"} and 50e9 < c4["kernels"]["k_format_chunks"]["write_bytes"] < 53e9   # 51 GB of lines
    line = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    assert line["parity_checked_paths"] > 1000 and line["config4"]["size"] == "full" and line["config4"]["lf_steps"] > 5e9
    for k in ("secondary", "high_degree", "search", "config4", "config4_small"):
        assert line[k]["cpu_baseline"]["kind"] == "port", k
    # every traffic file the committed line cites is a committed file of the same build
    cited = set(re.findall(r"profiles/[A-Za-z0-9_]+hbm_traffic\.json", open(os.path.join(ROOT, "profiles", "r05_bench.json")).read()))
    assert len(cited) == 6 and all(os.path.exists(os.path.join(ROOT, c)) for c in cited), cited
    assert line["roofline"]["source_fingerprint"] in prints


def test_round6_profiles_cover_every_config():
    """Round 6: the same set of PMC traffic files, of ONE build; the committed bench line has the shape this round's changes give it -- the
    headline handle opened for extraction only, the cpu_baseline timed on the plain walk (checksum walk untimed: its ratio on the line),
    config 4's first request next to its steady passes with the line sizes found at open, search calls into kept result arrays."""
    keys = {"r06_hbm_traffic.json": "sites=333334 haplotypes=5000 model=mosaic seed=42", "r06_secondary_hbm_traffic.json": "secondary",
            "r06_high_degree_hbm_traffic.json": "high_degree", "r06_search_hbm_traffic.json": "search", "r06_config4_hbm_traffic.json": "config4",
            "r06_config4_small_hbm_traffic.json": "config4_small"}
    prints = set()
    for name, key in keys.items():
        t = json.load(open(os.path.join(ROOT, "profiles", name)))
        factor = t.get("fetch_factor", 2.0)
        assert t["workload_key"] == key and len(t["source_fingerprint"]) == 16 and factor == (1.0 if key == "search" else 2.0), name
        assert t["traffic_bytes_per_launch"] == factor * t["fetch_bytes_raw"] + t["write_bytes"] and t["kernels"], name
        prints.add(t["source_fingerprint"])
    assert len(prints) == 1
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    assert line["open"]["flags"] == "GBWT_HIP_OPEN_EXTRACT" and line["open"]["open_ms"] < 30 and line["parity_checked_paths"] > 1000
    assert 0.8 < line["cpu_baseline"]["checksum_walk_ratio"] < 2.0 and line["cpu_baseline"]["kind"] == "port"
    c4 = line["config4"]
    assert c4["size"] == "full" and c4["line_sizes_ms"] > 0 and c4["walk_format"]["first_request_ms"] < 1.25 * c4["walk_format"]["ms"] + 10
    assert c4["kernel"] == "k_walk_direct + k_format_chunks" and c4["value_first_request"] > 0
    for form in ("unidirectional", "bidirectional"):
        assert line["search"][form]["wall_ms_reused_results"] <= line["search"][form]["wall_ms"] * 1.2
    cited = set(re.findall(r"profiles/[A-Za-z0-9_]+hbm_traffic\.json", open(os.path.join(ROOT, "profiles", "r06_bench.json")).read()))
    assert len(cited) == 6 and all(os.path.exists(os.path.join(ROOT, c)) for c in cited), cited
    assert line["roofline"]["source_fingerprint"] in prints


def test_gpus_without_a_launcher_starts_the_ranks(monkeypatch):
    """--gpus N > 1 outside torchrun: the ranks are children of this (GPU-free) process, started over 127.0.0.1."""
    calls = []
    import subprocess
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_must_match_gpus(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value)
