#!/bin/bash
# r06: the binary's 10 M-query run phase by phase (input read beside the index open), and what one output file in /dev/shm takes
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
nproc
g++ -O2 -std=c++17 -pthread tools/write_bench.cpp -o /tmp/write_bench && /tmp/write_bench /dev/shm/wb.out 5000 64 2>&1 | tee $O/write_bench.txt
timeout 900 python bench.py --keep-index --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --parity-queries 0 --cli-queries 0 > $O/c14_base.line 2> $O/c14_base.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 900 python tools/cli_10m.py $FM9 10000000 3 2>&1 | tee $O/cli_10m.txt
rm -f /dev/shm/dicey_bench_*
