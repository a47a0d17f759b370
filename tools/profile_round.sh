#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench, PMC passes, gather microbenchmark.
# Everything lands in gpurun_out/; the summaries worth judging are copied to profiles/ afterwards.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
R=${1:-r01}
# 1. the bench line itself (un-profiled)
timeout 900 python bench.py --keep-index > $OUT/bench_$R.json 2> $OUT/bench_$R.err
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
echo "index: $FM9" > $OUT/notes.txt
# 2. kernel trace + stats of the same command (index reused so the trace holds the search path, not the builder)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_traced.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
# 3. PMC passes (separate runs, counters only)
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  N=$(echo $C | tr ' ' '_')
  (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_$N -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/pmc_$N.json 2> $GRAFT_REPO_ROOT/$OUT/pmc_$N.err)
done
# 4. gather microbenchmark: ceiling + FETCH_SIZE calibration on a known byte count
tools/microbench/gather_bench 4 256 1 200000 > $OUT/gather_dep.txt 2>&1
tools/microbench/gather_bench 4 256 0 200000 > $OUT/gather_indep.txt 2>&1
tools/microbench/gather_bench 4 256 1 2000000 > $OUT/gather_dep_2M.txt 2>&1
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pmc_gather -o pmc --output-format csv -- $GRAFT_REPO_ROOT/tools/microbench/gather_bench 4 256 1 200000 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_gather.err)
rm -f /dev/shm/dicey_bench_*
find $OUT -name "*.csv" | head -50
du -sh $OUT
