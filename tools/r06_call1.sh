#!/bin/bash
# r06, first GPU call: the tests that cover what changed (locate job lists, context records, padlock chunks, ranks mode), then the
# repeats-genome configuration un-profiled, its kernel stats, and a dump of one batch's locate jobs.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py tests/test_gpu_padlock.py::test_padlock_scan_with_one_exon_longer_than_a_threads_share \
  tests/test_gpu_multirank.py::test_cli_ranks_answer_refused_chunks_in_pieces_and_fail_together -m gpu -x -q --durations=8 > $O/pytest1.log 2>&1
tail -15 $O/pytest1.log
timeout 900 python bench.py --genome repeats --steps 30 --warmup 8 --no-extras --no-extra-configs --cpu-seconds 3 --parity-queries 300 --keep-index \
  --detail-out $O/repeats_detail.json > $O/repeats.json 2> $O/repeats.err
tail -1 $O/repeats.json | cut -c1-600
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/repeats_detail.json"))
print("repeats:", d["value"], d["ms_per_step"], d["phases_ms"], d.get("parity_sample"), d.get("value_one_in_flight", {}).get("value"))
PY
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_rep -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --genome repeats --fm9 $FM9 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 --steps 12 --warmup 4 --detail-out $GRAFT_REPO_ROOT/$O/rep_traced_detail.json > $GRAFT_REPO_ROOT/$O/rep_traced.json 2> $GRAFT_REPO_ROOT/$O/rep_trace.err)
python - <<'PY'
import csv, sys, os
sys.path.insert(0, "tools")
from profnames import short_kernel_name
rows = list(csv.DictReader(open("gpurun_out/r06/trace_rep/trace_kernel_stats.csv")))
with open("gpurun_out/r06/repeats_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_ms,avg_us,min_us,max_us\n")
    for r in rows:
        n = short_kernel_name(r["Name"])
        f.write('"%s",%s,%.3f,%.2f,%.2f,%.2f\n' % (n[:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
        if int(r["Calls"]) >= 5 and "dg::" in n:
            print("%-46s calls %4s avg %9.1f min %9.1f max %9.1f us" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
DICEY_DUMP_JOBS=$GRAFT_REPO_ROOT/$O/jobs.bin timeout 300 python bench.py --genome repeats --fm9 $FM9 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 --steps 2 --warmup 1 --in-flight 1 --detail-out /tmp/x.json > /dev/null 2> $O/dump.err
ls -la $O/jobs.bin
find $O -name "*.db" -delete; rm -rf $O/trace_rep/*/*.db
rm -f /dev/shm/dicey_bench_*
