#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05/pytest_o.log 2>&1
tail -3 gpurun_out/r05/pytest_o.log
bash tools/r05_exp.sh 8 tools/r05_exp8.list 2>&1 | grep -v "   pmc"
