#!/bin/bash
# r04 call 4: PREP form of k_search1s (prepare + take inside), 8 wavefronts per SIMD again; parity subset, bench, kernel stats,
# repeats genome kernel stats.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capped.py tests/test_gpu_fullsize_layout.py -x -q > gpurun_out/r04/pytest_gpu3.log 2>&1
tail -6 gpurun_out/r04/pytest_gpu3.log
timeout 900 python bench.py --no-extra-configs --steps 20 --keep-index > gpurun_out/r04/bench_d.json 2> gpurun_out/r04/bench_d.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_d.json') if l.startswith('{')][-1])
print('value', j['value'], j['ms_per_step'], j['phases_ms']); print('same', j.get('value_same_batch')); print('h2h', j.get('host_to_host_pipelined')); print('d2h', j.get('value_with_d2h')); print('cli', j.get('cli_end_to_end')); print('parity', j.get('parity_sample'))
PY
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
bash tools/kstats.sh r04d --fm9 $FM9 --no-extra-configs --steps 20
DICEY_EXP_NOLEAVE=1 timeout 600 python bench.py --fm9 $FM9 --no-extra-configs --no-cpu-baseline --no-extras --steps 20 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('noleave', j['value'], j['phases_ms'])"
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --genome repeats --no-extra-configs --no-extras --steps 5 --cpu-seconds 3 --keep-index > gpurun_out/r04/bench_rep.json 2> gpurun_out/r04/bench_rep.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_rep.json') if l.startswith('{')][-1])
print('repeats', j['value'], j['ms_per_step'], j['phases_ms'], j.get('parity_sample'))
PY
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
bash tools/kstats.sh r04rep --fm9 $FM9 --genome repeats --no-extra-configs --steps 5
rm -f /dev/shm/dicey_bench_*
