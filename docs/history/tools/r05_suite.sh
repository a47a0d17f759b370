#!/bin/bash
# GPU box: the whole -m gpu suite with per-test durations, then the driver-shaped bench run.  usage: tools/r05_suite.sh <tag>
set -u
cd "$GRAFT_REPO_ROOT"
T=${1:-a}
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > gpurun_out/r05/pytest_$T.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05/pytest_$T.log
tail -3 gpurun_out/r05/pytest_$T.log
timeout 900 python bench.py --steps 20 --warmup 5 --detail-out gpurun_out/r05/bench_detail_$T.json > gpurun_out/r05/bench_$T.json 2> gpurun_out/r05/bench_$T.err
echo "bench rc=$?"
tail -c 4200 gpurun_out/r05/bench_$T.json
