#!/bin/bash
# r06: the binary's 10 M-query run: leaving with _exit, and without the preceding-characters array
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python bench.py --keep-index --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --parity-queries 0 --cli-queries 0 > $O/c16_base.line 2> $O/c16_base.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
V=$GRAFT_REPO_ROOT/dicey_amd/variants/libdiceygpu_exp.so
(echo "== unwinding exit"; timeout 900 python tools/cli_10m.py $FM9 10000000 2 DICEY_NO_QUICK_EXIT=1
 echo "== quick exit"; timeout 900 python tools/cli_10m.py $FM9 10000000 3
 echo "== quick exit, development library, as is"; timeout 900 python tools/cli_10m.py $FM9 10000000 2 LD_PRELOAD=$V
 echo "== quick exit, development library, no pre5"; timeout 900 python tools/cli_10m.py $FM9 10000000 3 LD_PRELOAD=$V DICEY_NO_PRE5=1) 2>&1 | grep -v amdgpu.ids | tee $O/cli_10m_b.txt
rm -f /dev/shm/dicey_bench_*
