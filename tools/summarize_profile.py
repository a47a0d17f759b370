#!/usr/bin/env python3
"""Turns gpurun_out/prof (tools/profile_round.sh) into the committed summaries under profiles/:
<round>_bench.json, <round>_bench_kernel_stats.csv, <round>_pmc_summary.csv, <round>_gather_bench.jsonl and
traffic_k_search.json (read by bench.py for roofline.traffic)."""
import csv, glob, json, os, re, shutil, sys
R = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, OUT = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
line = [l for l in open(os.path.join(P, f"bench_{R}.json")) if l.startswith("{")][-1]
open(os.path.join(OUT, f"{R}_bench.json"), "w").write(line)
bench = json.loads(line)
shutil.copy(os.path.join(P, "trace", "trace_kernel_stats.csv"), os.path.join(OUT, f"{R}_bench_kernel_stats.csv"))
rows, means = [], {}
for d in sorted(glob.glob(os.path.join(P, "pmc_*", "pmc_counter_collection.csv"))):
    acc = {}
    for r in csv.DictReader(open(d)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "dg::" not in k and "gather" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        rows.append((k, c, len(v), sum(v) / len(v), min(v), max(v)))
        means[(k, c)] = sum(v) / len(v)
with open(os.path.join(OUT, f"{R}_pmc_summary.csv"), "w") as f:
    f.write(f"# {R} PMC summary (rocprofv3 --pmc, one counter set per pass; bench.py --steps 2 --warmup 1; {bench['config']['workload']})\n")
    f.write("# per-dispatch means; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them\nkernel,counter,dispatches,mean,min,max\n")
    for r in rows: f.write("%s,%s,%d,%g,%g,%g\n" % r)
ks = [k for (k, c) in means if "k_search" in k][0]
gk = [k for (k, c) in means if "gather" in k]
fetch, write = means[(ks, "FETCH_SIZE")] * 1024, means[(ks, "WRITE_SIZE")] * 1024
old = json.load(open(os.path.join(OUT, "traffic_k_search.json")))
cal = old["fetch_size_calibration"]
if gk and (gk[0], "FETCH_SIZE") in means:
    cal = dict(cal, FETCH_SIZE_bytes=means[(gk[0], "FETCH_SIZE")] * 1024)
    cal["factor"] = cal["FETCH_SIZE_bytes"] / cal["known_bytes"]
t = {"workload": old["workload"], "kernel": "k_search<true,1>", "round": R, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
     "hbm_bytes_per_launch": fetch / cal["factor"] + write, "fetch_size_calibration": cal,
     "tcc": {"hit": means.get((ks, "TCC_HIT_sum")), "miss": means.get((ks, "TCC_MISS_sum")), "ea_rdreq": means.get((ks, "TCC_EA0_RDREQ_sum"))}}
json.dump(t, open(os.path.join(OUT, "traffic_k_search.json"), "w"), indent=1)
with open(os.path.join(OUT, f"{R}_gather_bench.jsonl"), "w") as f:
    for n in ("gather_dep.txt", "gather_indep.txt", "gather_dep_2M.txt"):
        p = os.path.join(P, n)
        if os.path.exists(p): f.write(json.dumps({"run": n, "output": open(p).read().strip().split("\n")}) + "\n")
print(json.dumps(t)[:400])
