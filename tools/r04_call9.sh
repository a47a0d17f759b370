#!/bin/bash
# r04 call 9: the round's evidence — profile round (bench line, kernel trace, PMC passes of the rotating run), new tests, repeats
# kernel stats, randomised differential runs.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
bash tools/profile_round.sh r04 > gpurun_out/r04/profile_round.log 2>&1
tail -c 400 gpurun_out/prof/bench_r04.json; echo
timeout 1500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_locate_topk.py "tests/test_gpu_parity.py::test_every_search_mode_gives_the_same_hits" tests/test_gpu_multirank.py -x -q > gpurun_out/r04/pytest_gpu6.log 2>&1
tail -4 gpurun_out/r04/pytest_gpu6.log
timeout 900 python bench.py --genome repeats --no-extra-configs --no-extras --steps 5 --cpu-seconds 3 --keep-index > gpurun_out/r04/bench_rep2.json 2> gpurun_out/r04/bench_rep2.err
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
bash tools/kstats.sh r04rep2 --fm9 $FM9 --genome repeats --no-extra-configs --steps 5 | head -20
rm -f /dev/shm/dicey_bench_*
export FUZZ_FAST_NEIGHBORS=1
for s in 1 2 3; do timeout 600 python tools/fuzz_hunt.py $s 40 > gpurun_out/r04/fuzz_hunt_$s.log 2>&1; tail -1 gpurun_out/r04/fuzz_hunt_$s.log; done
for s in 1 2; do timeout 600 python tools/fuzz_search.py $s 25 > gpurun_out/r04/fuzz_search_$s.log 2>&1; tail -1 gpurun_out/r04/fuzz_search_$s.log; done
for s in 1 2; do timeout 600 python tools/fuzz_padlock.py $s 25 > gpurun_out/r04/fuzz_padlock_$s.log 2>&1; tail -1 gpurun_out/r04/fuzz_padlock_$s.log; done
