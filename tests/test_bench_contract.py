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


# ---- r06: the five evidence fixes the r05 verdict asked for ("Next round" item 3) -------------------------------------------------
def test_profile_keys_tell_hamming_and_n_mix_apart():
    """(i) a sub-line may quote PMC traffic only from a profile of ITS workload: file name and workload tag carry the mode"""
    import bench
    edit2 = (bench.traffic_file(2, False, 0.0, "iid"), bench.workload_tag(100000, 20, 2, False, 0.0, 3.1e9, "iid"))
    ham2 = (bench.traffic_file(2, True, 0.0, "iid"), bench.workload_tag(100000, 20, 2, True, 0.0, 3.1e9, "iid"))
    nmix = (bench.traffic_file(1, False, 0.05, "iid"), bench.workload_tag(100000, 20, 1, False, 0.05, 3.1e9, "iid"))
    d1 = (bench.traffic_file(1, False, 0.0, "iid"), bench.workload_tag(100000, 20, 1, False, 0.0, 3.1e9, "iid"))
    rep = (bench.traffic_file(1, False, 0.0, "repeats"), bench.workload_tag(100000, 20, 1, False, 0.0, 3.1e9, "repeats"))
    assert len({x[0] for x in (edit2, ham2, nmix, d1, rep)}) == 5 and len({x[1] for x in (edit2, ham2, nmix, d1, rep)}) == 5
    assert d1 == ("traffic_k_search.json", "100000x20mer_d1_n3100000000_iid")          # the committed r05 files keep their names
    assert edit2[0] == "traffic_k_search_d2_iid.json" and rep[0] == "traffic_k_search_d1_repeats.json"
    # the edit-distance-2 profile is NOT what a Hamming-distance-2 line finds, whatever build it was taken on
    prof = os.path.join(ROOT, "profiles", edit2[0])
    if os.path.exists(prof):
        j = json.load(open(prof))
        assert j["workload"] == edit2[1] and j["workload"] != ham2[1]


def _line_with(extra):
    import bench
    out = {"metric": bench.METRIC, "value": 5.0e8, "unit": "primers/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.2, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": {"workload": "x", "genome": "y", "queries_per_gpu": 100000},
           "roofline": {"bound": "hbm", "kernel": "k_search1s<true, true>", "achieved": 800.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None,
                        "algorithmic_bytes_per_launch": 1.2e8, "kernel_ms": 0.15,
                        "kernel_ms_is": "busy time per launch: union of the timed launches' intervals (HIP events on the lanes' common timeline) / launches"},
           "cpu_baseline": {"value": 1366.0, "unit": "primers/s", "cores": 1, "kind": "port", "sample": "s"}, "parity_sample": None}
    out.update(extra)
    return _strict(bench.contract_line(bench._finite(out), "d.json"))


def test_contract_line_carries_sustained_and_the_parallel_cpu_figure():
    """(ii) + (iii)"""
    j = _line_with({"sustained": {"value": 5.1e8, "unit": "primers/s", "seconds": 1.18, "steps": 6000, "ms_per_step": 0.196, "note": "n" * 500},
                    "cpu_baseline_parallel": {"value": 18100.0, "unit": "primers/s", "cores": 128, "kind": "port", "cpu_model": "m", "sample": "s" * 900}})
    assert j["sustained"] == {"value": 5.1e8, "unit": "primers/s", "seconds": 1.18, "steps": 6000, "ms_per_step": 0.196}
    assert j["sustained"]["seconds"] >= 1.0
    assert j["cpu_baseline_parallel"] == {"value": 18100.0, "unit": "primers/s", "cores": 128, "kind": "port"}
    assert j["cpu_baseline"]["cores"] == 1
    j0 = _line_with({})
    assert j0["sustained"] is None and j0["cpu_baseline_parallel"] is None   # the keys are always there


def test_roofline_quotes_the_rocprof_average_next_to_the_busy_time():
    """(iv) kernel_avg_us_rocprof / frac_rocprof_avg come from profiles/kernel_stats_<tag>.json of the same build and workload, else null"""
    import bench
    roof = {"bound": "hbm", "kernel": "k_search1s<true, true>", "achieved": 800.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1,
            "algorithmic_bytes_per_launch": 122.9e6, "kernel_ms": 0.1547, "kernel_ms_is": "busy time per launch: union of ..."}
    tag = "_test_tag"
    prof = os.path.join(ROOT, "profiles", bench.kernel_stats_file(tag))
    try:
        json.dump({"build_id": bench.build_id(), "workload": tag,
                   "kernels": {"k_search1s<true, true>": {"calls": 47, "avg_us": 195.6, "min_us": 139.4, "max_us": 425.8}}}, open(prof, "w"))
        r = bench.rocprof_average(roof, tag)
        assert r["kernel_avg_us_rocprof"] == 195.6 and r["kernel_ms"] == 0.1547
        assert abs(r["frac_rocprof_avg"] - 122.9e6 / 195.6e-6 / 8e12) < 1e-9 and r["frac_rocprof_avg"] < r["frac"]
        j = _line_with({"roofline": r})
        assert j["roofline"]["kernel_avg_us_rocprof"] == 195.6 and abs(j["roofline"]["frac_rocprof_avg"] - 0.0785) < 1e-3
        assert j["roofline"]["kernel_ms_is"].startswith("busy time")
        assert bench.rocprof_average(dict(roof, kernel="k_search1s<true, false>"), tag)["kernel_avg_us_rocprof"] is None   # another instantiation
        assert bench.rocprof_average(roof, "other_tag")["kernel_avg_us_rocprof"] is None                                  # another workload
        json.dump({"build_id": "0" * 16, "workload": tag, "kernels": {"k_search1s<true, true>": {"calls": 1, "avg_us": 1.0}}}, open(prof, "w"))
        assert bench.rocprof_average(roof, tag)["kernel_avg_us_rocprof"] is None                                          # another build
    finally:
        os.remove(prof)


def test_cap_enum_block_is_labelled_latency_bound():
    """(v) no fraction of the HBM peak is claimed for k_cap_enum"""
    import bench
    r = bench.cap_enum_roofline(4000, 25, 2, 18.3)
    assert r["bound"] == "latency" and r["frac"] is None and r["kernel"] == "k_cap_enum" and r["traffic"] is None
    assert r["achieved"] > 0 and 0 < r["hbm_frac_upper_bound"] < 1
    j = _line_with({"roofline": bench.rocprof_average(r, "none")})
    assert j["roofline"]["bound"] == "latency" and j["roofline"]["frac"] is None and j["roofline"]["frac_rocprof_avg"] is None
