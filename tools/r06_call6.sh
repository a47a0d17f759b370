#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
T=${1:-d}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q > $O/pytest_$T.log 2>&1
tail -4 $O/pytest_$T.log
DICEY_TIMING=1 timeout 900 python bench.py --genome repeats --steps 30 --warmup 8 --no-extras --no-extra-configs --cpu-seconds 3 --parity-queries 300 --keep-index \
  --detail-out $O/repeats_detail_$T.json > $O/repeats_$T.json 2> $O/repeats_$T.err
grep "dicey timing" $O/repeats_$T.err | tail -22
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
python - $T <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06/repeats_detail_%s.json" % sys.argv[1]))
print("repeats:", "%.1f M" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in d["phases_ms"].items()}, d.get("parity_sample"),
      "one at a time %.1f M" % (d.get("value_one_in_flight", {}).get("value", 0) / 1e6), "hbm GB %.1f" % (d["index"]["hbm_bytes"] / 1e9))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_rep_$T -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --genome repeats --fm9 $FM9 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 --steps 12 --warmup 4 --detail-out $GRAFT_REPO_ROOT/$O/rep_traced_detail_$T.json > $GRAFT_REPO_ROOT/$O/rep_traced_$T.json 2> $GRAFT_REPO_ROOT/$O/rep_trace_$T.err)
python - $T <<'PY'
import csv, sys, os
sys.path.insert(0, "tools")
from profnames import short_kernel_name
T = sys.argv[1]
rows = list(csv.DictReader(open("gpurun_out/r06/trace_rep_%s/trace_kernel_stats.csv" % T)))
with open("gpurun_out/r06/repeats_kernel_stats_%s.csv" % T, "w") as f:
    f.write("kernel,calls,total_ms,avg_us,min_us,max_us\n")
    for r in rows:
        n = short_kernel_name(r["Name"])
        f.write('"%s",%s,%.3f,%.2f,%.2f,%.2f\n' % (n[:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
        if "dg::" in n and float(r["AverageNs"]) > 15e3:
            print("%-46s calls %4s avg %9.1f min %9.1f max %9.1f us" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $O/trace_rep_$T/*/*.db; find $O -name "*.db" -delete
rm -f /dev/shm/dicey_bench_*
