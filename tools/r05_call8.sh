#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_layout.py -x -q --durations=5 > gpurun_out/r05/pytest_k.log 2>&1
tail -8 gpurun_out/r05/pytest_k.log
bash tools/r05_exp.sh 6 tools/r05_exp6.list 2>&1 | grep -v "k_search<true, 1>\|k_search1s<true, false>"
