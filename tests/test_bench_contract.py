"""bench.py's contract line (what the driver parses): compact form of a full detail object, size limit, strict JSON, and the
interval arithmetic behind roofline.kernel_ms when several batches are in flight.  No GPU needed.

r04's driver record held no parsed line: bench.py printed one 25.6 KB JSON line and the driver keeps a bounded tail of stdout.  The
line is now < 4 KB (bench.CONTRACT_LINE_LIMIT), the detail object goes to a side file."""
import json
import math
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_union_of_intervals():
    import bench
    assert bench.union_of_intervals([]) == 0.0
    assert bench.union_of_intervals([(0.0, 1.0)]) == 1.0
    assert bench.union_of_intervals([(0.0, 1.0), (2.0, 3.5)]) == 2.5                      # apart
    assert bench.union_of_intervals([(0.0, 1.0), (0.5, 1.5), (1.2, 2.0)]) == 2.0          # a chain of overlaps
    assert bench.union_of_intervals([(2.0, 3.0), (0.0, 5.0), (1.0, 1.5)]) == 5.0          # one inside another, unsorted
    assert abs(bench.union_of_intervals([(0.0, 0.3), (0.2, 0.5), (0.45, 0.75)]) - 0.75) < 1e-12  # three lanes, every launch overlaps


def _strict(line):
    def bad(x):
        raise ValueError("not JSON: " + x)
    return json.loads(line, parse_constant=bad)


def check_contract(j, base):
    assert j["metric"] == base["metric"] and j["unit"] == "primers/s" and j["higher_is_better"] is True
    assert j["n_gpus"] >= 1 and j["steps"] >= 1 and j["warmup"] >= 0 and j["scaling"] in ("weak", "strong") and j["vs_baseline"] is None
    assert j["data"] == "synthetic" and j["dtype"] == "u32" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["ms_per_step"] > 0
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "traffic" in r and r["kernel"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]


def test_compact_line_of_the_r04_detail_object():
    """r04's 25.6 KB line is a full detail object: its compact form must carry the contract and fit"""
    import bench
    full = json.loads([ln for ln in open(os.path.join(ROOT, "profiles", "r04_final_bench.json")) if ln.startswith("{")][-1])
    assert len(json.dumps(full)) > 20000
    line = bench.contract_line(bench._finite(full), "bench_detail.json")
    assert len(line) < bench.CONTRACT_LINE_LIMIT == 4096 and "\n" not in line
    j = _strict(line)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    check_contract(j, base)
    assert abs(j["value"] - j["config"]["queries_per_gpu"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "primers/s" and c["sample"]
    assert j["parity_sample"]["mismatching"] == 0 and j["parity_sample"]["queries"] >= 1000
    for k in ("summary_hunt_d1_repeats", "summary_hunt_d2", "summary_hunt_d2_25mers", "summary_search", "summary_padlock"):
        assert j[k]["value"] > 0, k
    assert j["detail"] == "bench_detail.json"


def test_compact_line_survives_hostile_detail():
    """NaN / Infinity never reach the line (strict JSON), long strings are cut, and the limit holds whatever the detail object grows to"""
    import bench
    out = {"metric": bench.METRIC, "value": 1.0e8, "unit": "primers/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": "w" * 5000, "genome": "g" * 5000, "queries_per_gpu": 100000, "sharding": "s" * 900},
           "roofline": {"bound": "hbm", "kernel": "k" * 400, "achieved": 800.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": float("nan"),
                        "algorithmic_bytes_per_launch": 8.0e8, "kernel_ms": 1.0, "junk": ["x" * 100] * 500},
           "cpu_baseline": {"value": 1000.0, "unit": "primers/s", "cores": 1, "kind": "port", "cpu_model": "m" * 300, "sample": "s" * 3000},
           "parity_sample": {"queries": 1000, "mismatching": 0, "hits": 5, "seconds": float("inf")},
           "extra_configs": {"big": ["y" * 1000] * 200}}
    for i in range(8):
        out["summary_cfg%d" % i] = {"value": 1.0, "unit": "primers/s", "ms_per_step": 2.0, "dominant_kernel": "k" * 48, "bound": "hbm", "frac": 0.1,
                                    "parity": {"queries": 10, "mismatching": 0}}
    line = bench.contract_line(bench._finite(out), "d.json")
    assert len(line) < 4096
    j = _strict(line)
    assert j["roofline"]["traffic"] is None and j["value"] == 1.0e8 and "extra_configs" not in j
    with pytest.raises(ValueError):
        json.dumps({"x": float("nan")}, allow_nan=False)


def test_emit_writes_detail_and_prints_one_line(tmp_path, capsys):
    import bench
    out = {"metric": bench.METRIC, "value": 2.0, "unit": "primers/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": {"workload": "x", "genome": "y"},
           "roofline": {"bound": "hbm", "kernel": "k", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1 / 8000.0, "traffic": None,
                        "algorithmic_bytes_per_launch": 1e6, "kernel_ms": 1.0}, "cpu_baseline": None, "parity_sample": None,
           "big": {"nested": [math.nan, 1.5, {"deep": math.inf}]}}
    d = tmp_path / "detail.json"
    bench.emit(out, str(d))
    printed = capsys.readouterr().out
    assert printed.count("\n") == 1 and printed.startswith("{")
    j = _strict(printed)
    assert j["value"] == 2.0
    full = _strict(d.read_text())
    assert full["big"]["nested"] == [None, 1.5, {"deep": None}] and full["value"] == 2.0


def test_build_id_is_stable_and_tracks_the_kernel_sources(tmp_path, monkeypatch):
    import bench
    a = bench.build_id()
    assert a == bench.build_id() and len(a) == 16
    # a profile file is honoured only when its build_id equals this tree's and the named fields match exactly
    prof = os.path.join(ROOT, "profiles", "_test_traffic.json")
    try:
        json.dump({"build_id": a, "kernel": "k_search1s<true, true>", "workload": "w", "hbm_bytes_per_launch": 5}, open(prof, "w"))
        assert bench.profile_of_this_build("_test_traffic.json", kernel="k_search1s<true, true>", workload="w")["hbm_bytes_per_launch"] == 5
        assert bench.profile_of_this_build("_test_traffic.json", kernel="k_search1s<true, false>", workload="w") is None
        assert bench.profile_of_this_build("_test_traffic.json", kernel="k_search1s", workload="w") is None
        json.dump({"build_id": "0" * 16, "kernel": "k_search1s<true, true>", "workload": "w"}, open(prof, "w"))
        assert bench.profile_of_this_build("_test_traffic.json", kernel="k_search1s<true, true>", workload="w") is None      # stale
    finally:
        os.remove(prof)


def test_summarize_profile_matches_the_full_instantiation():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import summarize_profile as S
    means = {("dg::k_search1s<true, false>", "FETCH_SIZE"): 1.0, ("dg::k_search1s<true, true>", "FETCH_SIZE"): 2.0,
             ("dg::k_search2p<true>", "FETCH_SIZE"): 3.0}
    assert S.pick_kernel(means, "k_search1s<true, true>") == "dg::k_search1s<true, true>"
    assert S.pick_kernel(means, "k_search1s<true,false>") == "dg::k_search1s<true, false>"
    assert S.pick_kernel(means, "k_search2p<true>") == "dg::k_search2p<true>"
    with pytest.raises(SystemExit):
        S.pick_kernel(means, "k_search1s")
    with pytest.raises(SystemExit):
        S.pick_kernel(means, "k_search2p<false>")
