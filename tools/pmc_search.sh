#!/bin/bash
# GPU box: SQ / TCC / TCP counters of the default bench's kernels (k_search in particular), one counter set per pass.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py --keep-index --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/p$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --steps 2 --warmup 1 ${BENCH_ARGS:-} > $GRAFT_REPO_ROOT/$OUT/p$i.json 2> $GRAFT_REPO_ROOT/$OUT/p$i.err)
done <<LIST
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU
FETCH_SIZE GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum
LIST
rm -f /dev/shm/dicey_bench_*
python - <<'PY'
import csv, glob, re, os
acc = {}
for d in sorted(glob.glob("gpurun_out/pmc/p*/pmc_counter_collection.csv")):
    for r in csv.DictReader(open(d)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "dg::" not in k: continue
        acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open("gpurun_out/pmc/summary.csv", "w") as f:
    f.write("kernel,counter,dispatches,mean\n")
    for (k, c), v in sorted(acc.items()):
        f.write("%s,%s,%d,%g\n" % (k, c, len(v), sum(v) / len(v)))
PY
grep k_search $OUT/summary.csv
tail -3 $OUT/p1.err
