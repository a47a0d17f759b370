#!/bin/bash
# GPU box: kernel-trace stats and the memory counters of one bench configuration.  usage: tools/prof_cfg.sh <label> <bench args...>
# Reuses /dev/shm/dicey_bench_*.fm9 when present (build it first with bench.py --keep-index).
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
L=$1; shift
OUT=gpurun_out/prof_$L
rm -rf $OUT; mkdir -p $OUT
FM9=$(ls /dev/shm/dicey_bench_*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --parity-queries 0 --keep-index --detail-out /tmp/build_detail.json "$@" > $OUT/build.json 2> $OUT/build.err
  FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
fi
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --no-extras --parity-queries 0 --steps 5 --warmup 2 --detail-out $GRAFT_REPO_ROOT/$OUT/bench_detail.json "$@" > $GRAFT_REPO_ROOT/$OUT/bench_traced.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --no-extras --parity-queries 0 --steps 2 --warmup 1 --detail-out $GRAFT_REPO_ROOT/$OUT/pmc_detail_$i.json "$@" > $GRAFT_REPO_ROOT/$OUT/pmc_$i.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_$i.err)
done <<LIST
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
LIST
python - "$OUT" <<'PY'
import csv, glob, re, sys, os
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'tools'))
from profnames import short_kernel_name
out = sys.argv[1]
with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ms,avg_us\n")
    for r in csv.DictReader(open(os.path.join(out, "trace", "trace_kernel_stats.csv"))):
        n = short_kernel_name(r["Name"])
        if "dg::" not in n: continue
        f.write('"%s",%s,%.3f,%.2f\n' % (n[:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
acc = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(d)):
        k = short_kernel_name(r["Kernel_Name"])
        if "dg::" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    f.write("kernel,counter,dispatches,mean\n")
    for (k, c), v in sorted(acc.items()):
        f.write("%s,%s,%d,%g\n" % (k[:80], c, len(v), sum(v) / len(v)))
PY
rm -rf $OUT/trace/*/*.db $OUT/pmc_*/*/*.db 2>/dev/null
cat $OUT/kernel_stats.csv
grep -E "k_search|k_group_select|k_verify" $OUT/pmc_summary.csv
