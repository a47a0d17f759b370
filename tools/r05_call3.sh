#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_multirank.py "tests/test_gpu_parity.py::test_decreasing_offsets_fail_loudly_even_with_a_length_bound" tests/test_gpu_parity.py::test_randomised_configurations_against_oracle "tests/test_gpu_parity.py::test_every_search_mode_gives_the_same_hits" tests/test_gpu_capped.py -x -q --durations=12 > gpurun_out/r05/pytest_c.log 2>&1
tail -25 gpurun_out/r05/pytest_c.log
bash tools/r05_exp.sh 2 tools/r05_exp2.list 2>&1 | tee gpurun_out/r05/exp2.log
