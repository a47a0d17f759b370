#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for L in 6 12 13 24; do tools/microbench/gather_bench filter2 27 200000 $L 5; done > gpurun_out/r05/gather_filter2.jsonl 2>&1
tools/microbench/gather_bench filter 8 200000 12 1 4 >> gpurun_out/r05/gather_filter2.jsonl 2>&1
tail -12 gpurun_out/r05/gather_filter2.jsonl
bash tools/r05_suite.sh e
