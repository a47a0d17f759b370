#!/usr/bin/env python3
"""profiles/<round>_k_search_launches.json from the kernel trace of a profile round (gpurun_out/prof/trace/trace_kernel_trace.csv):
every k_search1s dispatch in start order with its duration and its overlap with other k_search1s dispatches, the launches that
ran alone against the ones that overlapped a neighbour (several batches in flight), and per stretch of overlapping launches the
union of their intervals — the time the kernel RAN — per launch.  usage: tools/summarize_launches.py <round> [kernel substring]"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_search1s"
rows = list(csv.DictReader(open(os.path.join(ROOT, "gpurun_out", "prof", "trace", "trace_kernel_trace.csv"))))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows if pat in r["Kernel_Name"])
out = []
for i, (s, e, q) in enumerate(ks):
    ov = sum(max(0, min(e, e2) - max(s, s2)) for j, (s2, e2, _) in enumerate(ks) if j != i)
    out.append({"start_us": round((s - ks[0][0]) / 1e3, 1), "duration_us": round((e - s) / 1e3, 2), "queue": q,
                "overlap_with_other_launches_us": round(ov / 1e3, 2)})
alone = [o["duration_us"] for o in out if o["overlap_with_other_launches_us"] < 1]
over = [o for o in out if o["overlap_with_other_launches_us"] >= 1]
iv = sorted((o["start_us"], o["start_us"] + o["duration_us"]) for o in over)
groups, cur = [], []
for a, b in iv:
    if cur and a - max(x[1] for x in cur) > 200:
        groups.append(cur)
        cur = []
    cur.append((a, b))
if cur:
    groups.append(cur)
stretches = []
for g in groups:
    union, (s, e) = 0.0, g[0]
    for a, b in g[1:]:
        if a <= e:
            e = max(e, b)
        else:
            union += e - s
            s, e = a, b
    union += e - s
    span = max(b for _, b in g) - g[0][0]
    stretches.append({"launches": len(g), "sum_of_durations_us": round(sum(b - a for a, b in g), 1), "union_of_busy_time_us": round(union, 1),
                      "busy_time_per_launch_us": round(union / len(g), 1), "span_us": round(span, 1), "kernel_running_fraction_of_span": round(union / span, 3)})
json.dump({"source": f"rocprofv3 --kernel-trace of `python bench.py --fm9 <index> --no-cpu-baseline --no-extras` (tools/profile_round.sh {rnd}): every {pat} dispatch in start order",
           "launches": out, "alone": {"n": len(alone), "avg_us": sum(alone) / max(1, len(alone))},
           "overlapping_another_launch": {"n": len(over), "avg_us": sum(o["duration_us"] for o in over) / max(1, len(over))},
           "overlapped_stretches": stretches,
           "note": "the timed region and its warm-up keep several batches in flight (one queue per lane): those launches overlap their neighbours; the passes behind the "
                   "timed region (stage times, value_same_batch, value_one_in_flight) run one batch at a time.  The stats CSV's average mixes both; the union of the "
                   "overlapping launches' intervals per launch is the kernel's busy time (bench.py: roofline.kernel_ms)"},
          open(os.path.join(ROOT, "profiles", f"{rnd}_k_search_launches.json"), "w"), indent=1)
print(json.dumps({"alone": len(alone), "alone_avg_us": sum(alone) / max(1, len(alone)), "overlapped": len(over), "stretches": stretches}))
