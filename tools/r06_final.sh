#!/bin/bash
# r06, the driver-shaped call: the whole GPU suite with durations, then `python bench.py --steps 20 --warmup 5`
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu_final.log 2>&1
tail -22 $O/pytest_gpu_final.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 --detail-out $O/final_bench_detail.json ) > $O/final_bench.json 2> $O/final_bench.err
tail -1 $O/final_bench.json | cut -c1-4000
tail -4 $O/final_bench.err
rm -f /dev/shm/dicey_bench_*
