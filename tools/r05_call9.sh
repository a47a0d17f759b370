#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_gpu_fullsize_layout.py tests/test_gpu_parity.py -x -q --durations=6 > gpurun_out/r05/pytest_n.log 2>&1
tail -12 gpurun_out/r05/pytest_n.log
bash tools/r05_exp.sh 7 tools/r05_exp7.list 2>&1 | grep -v "   pmc"
