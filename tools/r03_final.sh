#!/bin/bash
# r03 final: whole GPU suite, profile round of the default configuration (bench line, kernel trace, PMC passes), kernel stats of the
# repeats and d2 configurations, fp64 / LDS counters of the thal() wave kernels
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r03z
rm -rf $OUT; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
bash tools/profile_round.sh r03 > $OUT/profile_round.log 2>&1
tail -3 $OUT/profile_round.log
timeout 900 python bench.py --keep-index --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
bash tools/kstats.sh r03z_d2 --config hunt_d2 --fm9 $FM9 --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs > $OUT/kstats_d2.log 2>&1
bash tools/prof_thal.sh > $OUT/prof_thal.log 2>&1
tail -3 $OUT/prof_thal.log
rm -f /dev/shm/dicey_bench_*
bash tools/kstats.sh r03z_repeats --genome repeats --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs > $OUT/kstats_repeats.log 2>&1
cat $OUT/kstats_repeats.log | tail -25
rm -f /dev/shm/dicey_bench_*
