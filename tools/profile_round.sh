#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench line, kernel-trace stats of the same command, PMC passes for the
# dominant kernel.  Everything lands in gpurun_out/prof; tools/summarize_profile.py copies the summaries to profiles/.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
R=${1:-r02}
# 1. the bench line itself (un-profiled)
timeout 900 python bench.py --keep-index --no-extra-configs --detail-out $OUT/bench_detail_$R.json > $OUT/bench_$R.json 2> $OUT/bench_$R.err
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
echo "index: $FM9" > $OUT/notes.txt
# 2. kernel trace + stats of the same command (index reused so the trace holds the search path, not the builder)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --no-extras --detail-out $GRAFT_REPO_ROOT/$OUT/bench_detail_traced.json > $GRAFT_REPO_ROOT/$OUT/bench_traced.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
# 3. PMC passes (separate runs, counters only)
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --no-extras --steps 6 --warmup 3 --detail-out $GRAFT_REPO_ROOT/$OUT/pmc_detail_$i.json > $GRAFT_REPO_ROOT/$OUT/pmc_$i.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_$i.err)
done <<LIST
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum
LIST
# 4. gather microbenchmark: FETCH_SIZE calibration on a known byte count of random 64-B lines
if [ -x tools/microbench/gather_bench ]; then
  tools/microbench/gather_bench 4 256 1 200000 > $OUT/gather_dep.txt 2>&1
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_gather -o pmc --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/gather_bench 4 256 1 200000 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_gather.err)
fi
[ -n "${KEEP_INDEX:-}" ] || rm -f /dev/shm/dicey_bench_*
du -sh $OUT
