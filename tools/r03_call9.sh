#!/bin/bash
# r03 call 9: counters of k_verify_memo on the repeats genome (where does the 1.05 ms go), speculative fetch / host-to-host line on the default config
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r03k
rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py --genome repeats --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 --keep-index > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --genome repeats --fm9 $FM9 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/pmc_$i.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_$i.err)
done <<LIST
FETCH_SIZE WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE
LIST
python - "$OUT" <<'PY'
import csv, glob, re, sys, os
out = sys.argv[1]
acc = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(d)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "dg::" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    f.write("kernel,counter,dispatches,mean,max\n")
    for (k, c), v in sorted(acc.items()):
        f.write("%s,%s,%d,%g,%g\n" % (k[:80], c, len(v), sum(v) / len(v), max(v)))
PY
rm -rf $OUT/pmc_*/*/*.db 2>/dev/null
grep -E "k_verify_memo|k_locate_topk|k_leaf_alive" $OUT/pmc_summary.csv
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --no-extra-configs --cpu-seconds 3 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r03k/bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print(round(d["value"]), d["ms_per_step"], d.get("value_with_d2h"), d.get("host_to_host_pipelined"), d.get("parity_sample"))
PY
rocprofv3 --list-avail 2>/dev/null | grep -iE "F64|LDS" | head -40
