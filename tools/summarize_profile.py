#!/usr/bin/env python3
"""Turns a profile directory of the GPU box (tools/profile_round.sh -> gpurun_out/prof, tools/prof_cfg.sh -> gpurun_out/prof_<label>)
into the committed summaries under profiles/: <round>_bench.json (compact line), <round>_bench_detail.json, <round>_bench_kernel_stats.csv,
<round>_pmc_summary.csv and the traffic file bench.py reads for roofline.traffic (HBM bytes of the dominant kernel per launch:
FETCH_SIZE corrected by the calibration factor measured on a known byte count of random 64-byte lines + WRITE_SIZE).

usage: summarize_profile.py <round> [<profile dir> [<traffic file name>]]

The traffic row is the kernel the bench line names — the FULL template instantiation, compared as a string (r04 took the first row
whose name began like it and so filled the headline's roofline.traffic from k_search1s<true, false>, a warm-up dispatch).
The file is stamped with the build_id of the sources (bench.build_id) and bench.py carries its numbers only into lines of that build."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profnames import short_kernel_name  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bare(name: str) -> str:
    """`dg::k_search1s<true, true>` / `dg::(anonymous namespace)::k_x<1>` -> `k_search1s<true, true>` (spaces normalised)"""
    s = re.sub(r"\s+", " ", name.strip())
    s = re.sub(r"\(anonymous namespace\)::", "", s)
    s = re.sub(r"^(?:[A-Za-z_]\w*::)+", "", s)
    return re.sub(r"\s*,\s*", ", ", s)


def pick_kernel(means: dict, kernel: str, counter: str = "FETCH_SIZE") -> str:
    want = bare(kernel.split(" (")[0])
    rows = sorted({k for (k, c) in means if c == counter})
    hit = [k for k in rows if bare(k) == want]
    if len(hit) != 1:
        raise SystemExit(f"summarize_profile: the PMC summary holds {len(hit)} rows named exactly {want!r} (rows: {rows})")
    return hit[0]


def main():
    R = sys.argv[1]
    P = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof")
    tname = sys.argv[3] if len(sys.argv) > 3 else "traffic_k_search.json"
    OUT = os.path.join(ROOT, "profiles")
    detail_path = os.path.join(P, f"bench_detail_{R}.json")
    if not os.path.exists(detail_path):
        detail_path = os.path.join(P, "bench_detail.json")
    bench = json.load(open(detail_path))
    json.dump(bench, open(os.path.join(OUT, f"{R}_bench_detail.json"), "w"))
    lp = os.path.join(P, f"bench_{R}.json")
    if os.path.exists(lp):
        line = [ln for ln in open(lp) if ln.startswith("{")][-1]
        open(os.path.join(OUT, f"{R}_bench.json"), "w").write(line)
    import bench as B
    stats = os.path.join(P, "trace", "trace_kernel_stats.csv")
    # the workload the profile was taken on: the profiled run's own word (config.workload_tag, r06), else rebuilt from its text
    wtag = bench["config"].get("workload_tag")
    if not wtag:
        m0 = re.search(r"(\d+) synthetic (\d+)-mers per GPU, (edit|Hamming) distance (\d+)", bench["config"]["workload"])
        n0 = re.search(r"(\d+)% of the queries with one N", bench["config"]["workload"])
        wtag = B.workload_tag(int(m0.group(1)), int(m0.group(2)), int(m0.group(4)), m0.group(3) == "Hamming", int(n0.group(1)) / 100 if n0 else 0.0,
                              3.1e9, "repeats" if "planted repeat" in bench["config"]["genome"] else "iid")
    if os.path.exists(stats):
        # the same rows as JSON, keyed by the bare instantiation name and stamped with build_id + workload: bench.py quotes the
        # rocprofv3 AVERAGE duration of the roofline's kernel from it, next to its own busy-time figure
        kern = {}
        for r in csv.DictReader(open(stats)):
            kern[bare(short_kernel_name(r["Name"]))] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                                        "max_us": float(r["MaxNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
        traced = os.path.join(P, "bench_detail_traced.json")
        tb = json.load(open(traced)) if os.path.exists(traced) else bench
        json.dump({"workload": wtag, "round": R, "build_id": tb.get("build_id") or B.build_id(),
                   "command": "rocprofv3 --kernel-trace --stats -- python bench.py --fm9 <index> --no-cpu-baseline --no-extras" , "kernels": kern},
                  open(os.path.join(OUT, B.kernel_stats_file(wtag)), "w"), indent=1)
    if os.path.exists(stats):
        with open(os.path.join(OUT, f"{R}_bench_kernel_stats.csv"), "w") as f:  # long template names shortened, nothing dropped
            f.write("# rocprofv3 --kernel-trace --stats of the bench command on the reused index (--no-cpu-baseline --no-extras)\n")
            f.write("kernel,calls,total_ms,avg_us,min_us,max_us\n")
            for r in csv.DictReader(open(stats)):
                n = short_kernel_name(r["Name"])
                f.write('"%s",%s,%.3f,%.2f,%.2f,%.2f\n' % (n[:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                         float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    rows, means, ndisp = [], {}, {}
    for d in sorted(glob.glob(os.path.join(P, "pmc_*", "pmc_counter_collection.csv")) + glob.glob(os.path.join(P, "pmc_*", "*", "*counter_collection.csv"))):
        acc = {}
        for r in csv.DictReader(open(d)):
            k = short_kernel_name(r["Kernel_Name"])
            if "dg::" not in k and "gather" not in k:
                continue
            acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            rows.append((k, c, len(v), sum(v) / len(v), min(v), max(v)))
            means[(k, c)] = sum(v) / len(v)
            ndisp[(k, c)] = len(v)
    with open(os.path.join(OUT, f"{R}_pmc_summary.csv"), "w") as f:
        f.write(f"# {R} PMC summary (rocprofv3 --pmc, one counter set per pass; {bench['config']['workload']})\n")
        f.write("# per-dispatch means; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them\nkernel,counter,dispatches,mean,min,max\n")
        for r in rows:
            f.write('"%s",%s,%d,%g,%g,%g\n' % r)
    rf = bench.get("roofline_search") or bench["roofline"]
    kernel = rf["kernel"]
    # the step's dominant stage is another kernel than the search (repeat-rich genome: verify / locate): its traffic goes to a file of
    # its own, which bench.py reads for that block (traffic_stage_<workload>.json)
    if bench.get("roofline_search") and bench["roofline"].get("stage") in ("ms_locate", "ms_verify"):
        sk = bench["roofline"]["kernel"]
        try:
            sks = pick_kernel(means, sk)
            sf, sw_ = means[(sks, "FETCH_SIZE")] * 1024, means[(sks, "WRITE_SIZE")] * 1024
            st = {"workload": wtag, "kernel": sk, "pmc_row": sks, "dispatches_in_row": ndisp[(sks, "FETCH_SIZE")], "round": R,
                  "build_id": bench.get("build_id") or B.build_id(), "fetch_bytes_per_launch": sf, "write_bytes_per_launch": sw_,
                  "hbm_bytes_per_launch": sf + sw_, "stage": bench["roofline"]["stage"],
                  "tcc": {"hit": means.get((sks, "TCC_HIT_sum")), "miss": means.get((sks, "TCC_MISS_sum")), "ea_rdreq": means.get((sks, "TCC_EA0_RDREQ_sum")),
                          "req": means.get((sks, "TCC_REQ_sum"))},
                  "sq": {"insts_valu": means.get((sks, "SQ_INSTS_VALU")), "insts_salu": means.get((sks, "SQ_INSTS_SALU")), "waves": means.get((sks, "SQ_WAVES"))}}
            json.dump(st, open(os.path.join(OUT, "traffic_stage_" + wtag + ".json"), "w"), indent=1)
            print("stage:", json.dumps(st)[:500])
        except SystemExit as e:
            print("summarize_profile: no PMC row for the dominant stage's kernel:", e, file=sys.stderr)
    ks = pick_kernel(means, kernel)
    gk = [k for (k, c) in means if "gather" in k]
    fetch, write = means[(ks, "FETCH_SIZE")] * 1024, means[(ks, "WRITE_SIZE")] * 1024
    cal = {"pattern": "random 64-byte lines, 4 x dwordx4 per lane (tools/microbench/gather_bench 4 GiB, 200192 lanes x 256 x 2 lines)",
           "known_bytes": 6559891456, "FETCH_SIZE_bytes": 6555276245.333333, "factor": 0.9992964501474417, "measured_in": "r04d",
           "note": "MI355X_MICROARCH.md says wide coalesced streams read 0.5x; this gather pattern reads 1.00x, so no doubling is applied; "
                   "WRITE_SIZE is used as reported"}
    if gk and (gk[0], "FETCH_SIZE") in means:
        cal = dict(cal, FETCH_SIZE_bytes=means[(gk[0], "FETCH_SIZE")] * 1024, measured_in=R)
        cal["factor"] = cal["FETCH_SIZE_bytes"] / cal["known_bytes"]
    if len(sys.argv) <= 3 and bench["config"].get("traffic_file"):
        tname = bench["config"]["traffic_file"]
    t = {"workload": wtag, "kernel": kernel, "pmc_row": ks,
         "dispatches_in_row": ndisp[(ks, "FETCH_SIZE")], "round": R,
         "build_id": bench.get("build_id") or B.build_id(),  # the sources the PROFILED run was built from (bench.py stamps its detail)
         "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
         "hbm_bytes_per_launch": fetch / cal["factor"] + write, "fetch_size_calibration": cal,
         "tcc": {"hit": means.get((ks, "TCC_HIT_sum")), "miss": means.get((ks, "TCC_MISS_sum")), "ea_rdreq": means.get((ks, "TCC_EA0_RDREQ_sum")),
                 "req": means.get((ks, "TCC_REQ_sum")), "ea_rdreq_32B": means.get((ks, "TCC_EA0_RDREQ_32B_sum")),
                 "ea_rdreq_dram": means.get((ks, "TCC_EA0_RDREQ_DRAM_sum")), "ea_wrreq": means.get((ks, "TCC_EA0_WRREQ_sum")),
                 "ea_wrreq_64B": means.get((ks, "TCC_EA0_WRREQ_64B_sum"))},
         "sq": {"insts_valu": means.get((ks, "SQ_INSTS_VALU")), "insts_salu": means.get((ks, "SQ_INSTS_SALU")), "waves": means.get((ks, "SQ_WAVES")),
                "insts_vmem_rd": means.get((ks, "SQ_INSTS_VMEM_RD")), "insts_vmem_wr": means.get((ks, "SQ_INSTS_VMEM_WR"))},
         "distinct_batches": bench["config"].get("distinct_batches")}
    if t["build_id"] != B.build_id():
        print("summarize_profile: note: the profiled run was build %s, this tree is %s — bench.py will report traffic: null until re-profiled"
              % (t["build_id"], B.build_id()), file=sys.stderr)
    json.dump(t, open(os.path.join(OUT, tname), "w"), indent=1)
    print(json.dumps(t)[:900])


if __name__ == "__main__":
    main()
