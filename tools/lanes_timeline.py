#!/usr/bin/env python3
"""profiles/<round>_lanes_timeline.txt from the kernel trace of a profile round (gpurun_out/prof/trace/trace_kernel_trace.csv): the
kernels of the timed region's middle stretch in start order (start, end, duration, queue, short name), the longest stretch without a
resident search kernel, and the share of the stretch with one.  usage: tools/lanes_timeline.py <round> [trace csv] [search kernel substring]"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from profnames import short_kernel_name
rnd = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "prof", "trace", "trace_kernel_trace.csv")
pat = sys.argv[3] if len(sys.argv) > 3 else "k_search1s<true, true>"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short_kernel_name(r["Kernel_Name"])) for r in csv.DictReader(open(path))]
rows = sorted(r for r in rows if "dg::" in r[3])
srch = [r for r in rows if pat in r[3]]
# the stretch in which the search launches of THREE queues alternate (the timed region and the sustained pass; the passes with one
# batch at a time use one queue): its longest run, without the first and last three launches (the pipeline filling and draining)
multi = [len({x[2] for x in srch[max(0, i - 3):i + 4]}) >= 3 for i in range(len(srch))]
best, cur = (0, 0), None
for i, m_ in enumerate(multi + [False]):
    if m_ and cur is None:
        cur = i
    elif not m_ and cur is not None:
        if i - cur > best[1] - best[0]:
            best = (cur, i)
        cur = None
lo_i, hi_i = best[0] + 3, max(best[0] + 4, best[1] - 3)
a, b = srch[lo_i][0], srch[hi_i - 1][1]
win = [r for r in rows if r[0] >= a and r[1] <= b]
qn = {q: "q%d" % i for i, q in enumerate(sorted({r[2] for r in win}))}
iv = sorted((r[0], r[1]) for r in win if pat in r[3])
cov, gaps, cur = 0, [], iv[0]
for s, e in iv[1:]:
    if s > cur[1]:
        cov += cur[1] - cur[0]
        gaps.append((s - cur[1], cur[1]))
        cur = (s, e)
    else:
        cur = (cur[0], max(cur[1], e))
cov += cur[1] - cur[0]
span = iv[-1][1] - iv[0][0]
with open(os.path.join(ROOT, "profiles", rnd + "_lanes_timeline.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace of `python bench.py` (three batches in flight), the stretch where three lanes' searches alternate\n")
    f.write("# a %s kernel is resident during %.1f %% of the %.0f us stretch; gaps without one: %d, longest %.1f us, mean %.1f us\n" % (
        pat, 100.0 * cov / span, span / 1e3, len(gaps), max(g[0] for g in gaps) / 1e3 if gaps else 0, (sum(g[0] for g in gaps) / len(gaps) / 1e3) if gaps else 0))
    f.write("# start_us end_us duration_us hw_queue kernel   (profile round %s)\n" % rnd)
    for s, e, q, n in win[:160]:
        f.write("%9.1f %9.1f %7.1f  %s %s\n" % ((s - a) / 1e3, (e - a) / 1e3, (e - s) / 1e3, qn[q], n.replace("dg::", "")))
print(open(os.path.join(ROOT, "profiles", rnd + "_lanes_timeline.txt")).read()[:600])
