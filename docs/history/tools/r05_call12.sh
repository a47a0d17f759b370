#!/bin/bash
# kernel trace of the default bench (three batches in flight) under the environment given in $TRACE -> gpurun_out/r05/trace_$TAG
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index --detail-out /tmp/b.json > /dev/null 2> gpurun_out/r05/trace_build.err
  FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
fi
for T in $TAGS; do
  case $T in
    new) E="X=1";;
    old) E="DICEY_NO_SEARCH_STREAM=1";;
  esac
  (cd /tmp && env $E timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r05/trace_$T -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 --steps 12 --warmup 4 --detail-out /tmp/d.json > gpurun_out_trace_$T.log 2>&1; tail -2 gpurun_out_trace_$T.log | cut -c1-300)
  find gpurun_out/r05/trace_$T -name "*kernel_trace.csv" | head -2
done
