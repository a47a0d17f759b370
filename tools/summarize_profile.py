#!/usr/bin/env python3
"""Turns gpurun_out/prof (tools/profile_round.sh) into the committed summaries under profiles/:
<round>_bench.json, <round>_bench_kernel_stats.csv, <round>_pmc_summary.csv and traffic_k_search.json (read by bench.py for
roofline.traffic: HBM bytes of the dominant kernel per launch, FETCH_SIZE corrected by the calibration factor measured on a
known byte count of random 64-byte lines + WRITE_SIZE)."""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profnames import short_kernel_name
R = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, OUT = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
line = [l for l in open(os.path.join(P, f"bench_{R}.json")) if l.startswith("{")][-1]
open(os.path.join(OUT, f"{R}_bench.json"), "w").write(line)
bench = json.loads(line)
with open(os.path.join(OUT, f"{R}_bench_kernel_stats.csv"), "w") as f:  # long template names shortened, nothing dropped
    f.write("# rocprofv3 --kernel-trace --stats of `python bench.py --fm9 <index of the bench run> --no-cpu-baseline --no-extras`\n")
    f.write("kernel,calls,total_ms,avg_us,min_us,max_us\n")
    for r in csv.DictReader(open(os.path.join(P, "trace", "trace_kernel_stats.csv"))):
        n = short_kernel_name(r["Name"])
        f.write('"%s",%s,%.3f,%.2f,%.2f,%.2f\n' % (n[:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                 float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
rows, means = [], {}
for d in sorted(glob.glob(os.path.join(P, "pmc_*", "pmc_counter_collection.csv"))):
    acc = {}
    for r in csv.DictReader(open(d)):
        k = short_kernel_name(r["Kernel_Name"])
        if "dg::" not in k and "gather" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        rows.append((k, c, len(v), sum(v) / len(v), min(v), max(v)))
        means[(k, c)] = sum(v) / len(v)
with open(os.path.join(OUT, f"{R}_pmc_summary.csv"), "w") as f:
    f.write(f"# {R} PMC summary (rocprofv3 --pmc, one counter set per pass; bench.py --steps 2 --warmup 1; {bench['config']['workload']})\n")
    f.write("# per-dispatch means; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them\nkernel,counter,dispatches,mean,min,max\n")
    for r in rows: f.write("%s,%s,%d,%g,%g,%g\n" % r)
kernel = bench["roofline"]["kernel"]
ks = [k for (k, c) in means if kernel.split("<")[0] in k and c == "FETCH_SIZE"][0]
gk = [k for (k, c) in means if "gather" in k]
fetch, write = means[(ks, "FETCH_SIZE")] * 1024, means[(ks, "WRITE_SIZE")] * 1024
old = json.load(open(os.path.join(OUT, "traffic_k_search.json")))
cal = old["fetch_size_calibration"]
if gk and (gk[0], "FETCH_SIZE") in means:
    cal = dict(cal, FETCH_SIZE_bytes=means[(gk[0], "FETCH_SIZE")] * 1024)
    cal["factor"] = cal["FETCH_SIZE_bytes"] / cal["known_bytes"]
m = re.search(r"(\d+) synthetic (\d+)-mers per GPU, edit distance (\d+)", bench["config"]["workload"])
gen = "repeats" if "planted repeat" in bench["config"]["genome"] else "iid"
n = bench["index"]["n"] - 14  # 24 separators are part of n; bench.py keys on the requested size
t = {"workload": f"{m.group(1)}x{m.group(2)}mer_d{m.group(3)}_n3100000000_{gen}", "kernel": kernel, "round": R,
     "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
     "hbm_bytes_per_launch": fetch / cal["factor"] + write, "fetch_size_calibration": cal,
     "tcc": {"hit": means.get((ks, "TCC_HIT_sum")), "miss": means.get((ks, "TCC_MISS_sum")), "ea_rdreq": means.get((ks, "TCC_EA0_RDREQ_sum")),
             "req": means.get((ks, "TCC_REQ_sum")), "ea_rdreq_32B": means.get((ks, "TCC_EA0_RDREQ_32B_sum")),
             "ea_rdreq_dram": means.get((ks, "TCC_EA0_RDREQ_DRAM_sum")), "ea_wrreq": means.get((ks, "TCC_EA0_WRREQ_sum"))},
     "distinct_batches": bench["config"].get("distinct_batches")}
json.dump(t, open(os.path.join(OUT, "traffic_k_search.json"), "w"), indent=1)
print(json.dumps(t)[:600])
