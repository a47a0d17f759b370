#!/bin/bash
# r04 call 2 (GPU box): GPU suite on the r04 kernels (bit-plane verify, k_search1s with early wavefront exit, compact results, two
# lanes per handle), then the bench line with host timing, then kernel stats of the timed region.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu.log 2>&1
tail -5 gpurun_out/r04/pytest_gpu.log
DICEY_TIMING=2 timeout 900 python bench.py --no-extra-configs --steps 20 --keep-index > gpurun_out/r04/bench_b.json 2> gpurun_out/r04/bench_b.err
tail -c 600 gpurun_out/r04/bench_b.json
grep "dicey timing: batch" gpurun_out/r04/bench_b.err | tail -12
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
bash tools/kstats.sh r04b --fm9 $FM9 --no-extra-configs --steps 20
rm -f /dev/shm/dicey_bench_*
