#!/bin/bash
# thal-side changes: host-side phase times of dg_padlock_scan (DICEY_TIMING) on the bench's padlock configuration, the search configuration, tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index --detail-out /tmp/b.json > /dev/null 2> gpurun_out/r05/pad_build.err
  FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
fi
DICEY_TIMING=1 timeout 900 python bench.py --fm9 $FM9 --config padlock --no-cpu-baseline --no-extras --no-extra-configs --steps 3 --warmup 1 --detail-out gpurun_out/r05/padlock_timing.json > gpurun_out/r05/padlock_timing.line 2> gpurun_out/r05/padlock_timing.err
grep "dg_padlock_scan" gpurun_out/r05/padlock_timing.err | tail -6
python -c "import json; j=json.load(open('gpurun_out/r05/padlock_timing.json')); print('padlock', j['value'], j['ms_per_step'], j.get('parity_sample'))"
timeout 900 python bench.py --fm9 $FM9 --config search --no-cpu-baseline --no-extras --no-extra-configs --steps 3 --warmup 1 --detail-out gpurun_out/r05/search_timing.json > gpurun_out/r05/search_timing.line 2> gpurun_out/r05/search_timing.err
python -c "import json; j=json.load(open('gpurun_out/r05/search_timing.json')); print('search', j['value'], j['ms_per_step'], j.get('parity_sample'))"
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -x -q > gpurun_out/r05/pytest_k.log 2>&1; tail -4 gpurun_out/r05/pytest_k.log; fi
