"""bench.py's JSON line (the driver's contract) on the committed driver-shaped run, and the interval arithmetic behind
roofline.kernel_ms when several batches are in flight.  No GPU needed."""
import json
import os
import sys

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


def test_committed_bench_line_obeys_the_contract():
    line = [ln for ln in open(os.path.join(ROOT, "profiles", "r04_final_bench.json")) if ln.startswith("{")][-1]
    j = json.loads(line)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"] == base["metric"] and j["unit"] == "primers/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] >= 1 and j["warmup"] >= 1 and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["data"] == "synthetic" and j["dtype"] == "u32" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - j["config"]["queries_per_gpu"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    # achieved = algorithmic bytes per launch / the kernel's busy time per launch (union of the timed launches' intervals)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    if r.get("busy"):
        assert abs(r["kernel_ms"] - r["busy"]["union_ms"] / r["busy"]["launches"]) < 1e-9
        assert r["busy"]["union_ms"] <= r["busy"]["sum_of_durations_ms"] + 1e-9 and r["busy"]["launches"] == j["steps"]
        assert r["busy"]["union_ms"] <= j["steps"] * j["ms_per_step"] + 1e-6           # the kernel cannot run longer than the timed region
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "primers/s" and c["sample"]
    assert j["parity_sample"]["mismatching"] == 0 and j["parity_sample"]["queries"] >= 1000
    for k in ("summary_hunt_d1_repeats", "summary_hunt_d2", "summary_hunt_d2_25mers"):
        assert j[k]["parity"]["mismatching"] == 0, k
