#!/bin/bash
# GPU box experiment: what bounds k_probe1?  Kernel times of the split form with parts of the probe kernel switched off.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/expp; rm -rf $OUT; mkdir -p $OUT
FM9=$(ls /dev/shm/dicey_bench_*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --parity-queries 0 --keep-index > $OUT/build.json 2> $OUT/build.err
  FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
fi
for M in 0 1 2 3 4 5 7; do
  (cd /tmp && DICEY_FLAT1_SPLIT=1 DICEY_DBG_PROBE=$M timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/t$M -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --no-extras --parity-queries 0 --steps 5 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/t$M.err)
  echo "mode $M: $(grep -E 'k_probe1|k_finish1' $OUT/t$M/t_kernel_stats.csv | awk -F, '{print $1, $4}' | tr '\n' ' ')"
  rm -rf $OUT/t$M/*.db
done
