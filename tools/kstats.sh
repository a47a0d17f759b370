#!/bin/bash
# GPU box: kernel stats (rocprofv3 --kernel-trace --stats) of one bench.py configuration; prints the dg:: kernels of the batch
# usage: tools/kstats.sh <tag> <bench args...>
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/kstats_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/bench.err)
python - "$OUT" <<'PY'
import csv, re, sys, os
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'tools'))
from profnames import short_kernel_name
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/trace/trace_kernel_stats.csv")))
keep = []
for r in rows:
    n = short_kernel_name(r["Name"])
    keep.append((n[:70], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
with open(out + "/kernel_stats_short.csv", "w") as f:
    f.write("kernel,calls,total_ms,avg_us\n")
    for k in keep: f.write("%s,%d,%.3f,%.2f\n" % k)
for k in keep:
    if k[1] >= 3 and "dg::" in k[0]: print("%-50s calls %4d avg %10.1f us" % (k[0][:50], k[1], k[3]))
PY
