#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
echo "== old library: the extended test must fail"
DICEY_LIB=$GRAFT_REPO_ROOT/dicey_amd/variants/libdiceygpu_old.so timeout 600 python -m pytest tests/test_gpu_locate_topk.py -m gpu -x -q -k "short_queries_that_end_in_n" 2>&1 | tail -3 | cut -c1-300
echo "== this build"
timeout 600 python -m pytest tests/test_gpu_locate_topk.py -m gpu -x -q -k "short_queries_that_end_in_n" 2>&1 | tail -2
bash tools/r06_fuzz3.sh > /dev/null 2>&1
grep -E "##|failing|MISMATCH" $O/fuzz_n.txt | cut -c1-200
