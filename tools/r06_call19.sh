#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
echo "== old library: the new test must fail"
DICEY_LIB=$GRAFT_REPO_ROOT/dicey_amd/variants/libdiceygpu_old.so timeout 600 python -m pytest tests/test_gpu_locate_topk.py -m gpu -x -q -k "short_queries_that_end_in_n" 2>&1 | tail -4 | cut -c1-300
echo "== this build"
timeout 1500 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py tests/test_gpu_fullsize_layout.py tests/test_gpu_cli.py tests/test_gpu_padlock.py -m gpu -x -q > $O/pytest_c19.log 2>&1
tail -3 $O/pytest_c19.log
bash tools/r06_fuzz.sh 2>&1 | grep -E "##|failing|MISMATCH"
timeout 900 python bench.py --keep-index --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --parity-queries 0 --cli-queries 0 > $O/c19_base.line 2> $O/c19_base.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 900 python tools/cli_10m.py $FM9 1000000 3 DIST=2 2>&1 | grep -v amdgpu.ids | tee $O/cli_d2_1m.txt
rm -f /dev/shm/dicey_bench_*
