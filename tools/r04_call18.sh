#!/bin/bash
# r04 call 18: whole GPU suite on the final build; counters of the repeats genome's step (fabric requests of every kernel).
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_gpu_final6.log 2>&1
tail -3 gpurun_out/r04/pytest_gpu_final6.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
rm -f /dev/shm/dicey_bench_*
bash tools/prof_cfg.sh r04rep --config hunt_d1 --genome repeats --no-extra-configs > gpurun_out/r04/prof_rep.log 2>&1
grep -E "TCC_EA0_RDREQ_sum|kernel,calls" gpurun_out/prof_r04rep/pmc_summary.csv | head -30
head -14 gpurun_out/prof_r04rep/kernel_stats.csv
rm -f /dev/shm/dicey_bench_*
