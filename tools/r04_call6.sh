#!/bin/bash
# r04 call 6: capped-neighbourhood tests with k_explicit behind the presence filter, the 25-mer sub-line with its kernel stats,
# thal counters (search / padlock), distance-2 kernel stats + PMC.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_capped.py tests/test_gpu_padlock.py -x -q > gpurun_out/r04/pytest_gpu4.log 2>&1
tail -4 gpurun_out/r04/pytest_gpu4.log
timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --qlen 25 --queries 2000 --steps 3 --warmup 1 --cpu-seconds 4 --parity-queries 100 --no-extras --no-extra-configs > gpurun_out/r04/bench_25.json 2> gpurun_out/r04/bench_25.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_25.json') if l.startswith('{')][-1])
print('25mers', j['value'], j['ms_per_step'], j['phases_ms'], j.get('cap_stage'), j.get('parity_sample'), j['roofline'].get('kernel'), j['roofline'].get('frac'))
PY
bash tools/kstats.sh r04_25 --fm9 $FM9 --config hunt_d2 --qlen 25 --queries 2000 --steps 3 --warmup 1 --no-extra-configs
bash tools/prof_cfg.sh r04d2 --config hunt_d2 --no-extra-configs
bash tools/prof_thal.sh
rm -f /dev/shm/dicey_bench_*
