#!/bin/bash
# r04 call 8: k_cap_enum with the Bloom filter in LDS (25-mer line + kernel stats), then the whole GPU suite on this build.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 900 python bench.py --fm9 $FM9 --config hunt_d2 --qlen 25 --queries 2000 --steps 3 --warmup 1 --cpu-seconds 4 --parity-queries 100 --no-extras --no-extra-configs > gpurun_out/r04/bench_25c.json 2> gpurun_out/r04/bench_25c.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_25c.json') if l.startswith('{')][-1])
print('25mers', j['value'], j['ms_per_step'], j.get('cap_stage'), j.get('parity_sample'), j['roofline'].get('kernel'), j['roofline'].get('frac'))
PY
bash tools/kstats.sh r04_25c --fm9 $FM9 --config hunt_d2 --qlen 25 --queries 2000 --steps 3 --warmup 1 --no-extra-configs | grep -E "cap_enum|explicit|search2p"
timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-extra-configs 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'])"
rm -f /dev/shm/dicey_bench_*
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_full.log 2>&1
tail -6 gpurun_out/r04/pytest_gpu_full.log
