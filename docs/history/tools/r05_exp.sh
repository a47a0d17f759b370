#!/bin/bash
# GPU box: A/B runs of bench.py variants on ONE index (built once), optional PMC passes per variant.
# usage: tools/r05_exp.sh <tag> <file with one variant per line: name|env assignments|bench args|pmc(0/1)>
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
T=$1; LIST=$2
OUT=gpurun_out/r05/exp_$T
rm -rf $OUT; mkdir -p $OUT
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index --detail-out $OUT/build_detail.json > $OUT/build.json 2> $OUT/build.err
  FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
fi
echo "index $FM9"
while IFS='|' read -r NAME ENVS ARGS PMC; do
  [ -z "$NAME" ] && continue
  case "$NAME" in \#*) continue;; esac
  env $ENVS timeout 600 python bench.py --fm9 $FM9 --no-extras --no-extra-configs --no-cpu-baseline $ARGS --detail-out $OUT/$NAME.json > $OUT/$NAME.line 2> $OUT/$NAME.err
  python - "$OUT/$NAME.json" "$NAME" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    r = j.get("roofline_search") or j["roofline"]
    print("%-22s value %.4g %s  ms/step %.4f  kernel %s ms %.4f (launch %.4f)  ext %.3g tab %.3g probe %.4g  parity %s" % (
        sys.argv[2], j["value"], j["unit"], j["ms_per_step"], r.get("kernel"), r.get("kernel_ms") or 0, r.get("launch_ms") or 0,
        r.get("ext_steps_per_launch") or 0, r.get("table_reads_per_launch") or 0, r.get("filter_probes_per_launch") or 0, j.get("parity_sample")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  if [ "${PMC:-0}" = "1" ]; then
    i=0
    for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
      i=$((i+1))
      (cd /tmp && env $ENVS timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_${NAME}_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 $ARGS --steps 6 --warmup 3 --detail-out /tmp/pmc_detail.json > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_${NAME}_$i.err)
    done
    python - "$OUT" "$NAME" <<'PY'
import csv, glob, os, sys
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'tools'))
from profnames import short_kernel_name
out, name = sys.argv[1], sys.argv[2]
acc = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_%s_*" % name, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(d)):
        k = short_kernel_name(r["Kernel_Name"])
        if "dg::" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc_%s.csv" % name), "w") as f:
    f.write("kernel,counter,dispatches,mean\n")
    for (k, c), v in sorted(acc.items()):
        f.write('"%s",%s,%d,%g\n' % (k[:80], c, len(v), sum(v) / len(v)))
        if "k_search" in k: print("   pmc", k[:40], c, len(v), "%g" % (sum(v) / len(v)))
PY
    rm -rf $OUT/pmc_${NAME}_*/
  fi
done < "$LIST"
rm -f /dev/shm/dicey_bench_*
