#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for L in 6 12 15 24; do tools/microbench/gather_bench filter2 27 200000 $L 5; done > gpurun_out/r05/gather_filter2b.jsonl 2>&1
tail -20 gpurun_out/r05/gather_filter2b.jsonl | cut -c60-
timeout 1200 python -m pytest tests/test_gpu_thal_wave.py tests/test_gpu_search.py tests/test_gpu_padlock.py -x -q --durations=8 > gpurun_out/r05/pytest_f.log 2>&1
tail -15 gpurun_out/r05/pytest_f.log
bash tools/r05_exp.sh 3 tools/r05_exp3.list 2>&1 | tee gpurun_out/r05/exp3.log
