#!/bin/bash
# GPU box experiment: k_probe1 / k_finish1 times against the long filter's order and number of copies.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/expp2; rm -rf $OUT; mkdir -p $OUT
FM9=$(ls /dev/shm/dicey_bench_*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --parity-queries 0 --keep-index > $OUT/build.json 2> $OUT/build.err
  FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
fi
for V in "19 4" "19 1" "19 2" "18 4" "17 4" "18 1"; do
  set -- $V
  (cd /tmp && DICEY_FLAT1_SPLIT=1 DICEY_KMER_K2=$1 DICEY_KF_COPIES=$2 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/t$1_$2 -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --no-extras --parity-queries 0 --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/b$1_$2.json 2> $GRAFT_REPO_ROOT/$OUT/t$1_$2.err)
  python - <<PY
import csv, json
o = []
for r in csv.DictReader(open("$OUT/t$1_$2/t_kernel_stats.csv")):
    if "k_probe1" in r["Name"] or "k_finish1" in r["Name"]:
        o.append((r["Name"][10:19], round(float(r["AverageNs"]) / 1e3, 1)))
try:
    j = json.load(open("$OUT/b$1_$2.json")); r = j["roofline"]
    o.append(("tab", r["table_reads_per_launch"])); o.append(("ext", r["ext_steps_per_launch"]))
except Exception as e:
    o.append(("json", str(e)))
print("K2=$1 copies=$2", o)
PY
  rm -rf $OUT/t$1_$2/*.db
done
