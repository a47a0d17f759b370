#!/bin/bash
# r04 call 11: filtered intervals (FmView::pre5) in k_search1s — parity first, then the same-box A/B against DICEY_NO_PRE5.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_padlock.py tests/test_gpu_fullsize_layout.py tests/test_gpu_locate_topk.py tests/test_gpu_cli.py tests/test_gpu_multirank.py -x -q > gpurun_out/r04/pytest_gpu8.log 2>&1
tail -4 gpurun_out/r04/pytest_gpu8.log
timeout 600 python bench.py --no-extra-configs --no-extras --steps 30 --cpu-seconds 2 --keep-index > gpurun_out/r04/pre5_on.json 2>/dev/null
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
DICEY_NO_PRE5=1 timeout 600 python bench.py --fm9 $FM9 --no-extra-configs --no-cpu-baseline --no-extras --steps 30 > gpurun_out/r04/pre5_off.json 2>/dev/null
timeout 600 python bench.py --fm9 $FM9 --no-extra-configs --no-cpu-baseline --no-extras --steps 30 > gpurun_out/r04/pre5_on2.json 2>/dev/null
for f in pre5_on pre5_off pre5_on2; do python - $f <<'PY'
import json,sys
j=json.loads([l for l in open('gpurun_out/r04/%s.json'%sys.argv[1]) if l.startswith('{')][-1])
r=j['roofline']
print(sys.argv[1], round(j['value']/1e6,1), 'M/s', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['phases_ms'].items()}, 'ext', r['ext_steps_per_launch'], 'tab', r['table_reads_per_launch'], 'frac', round(r['frac'],3), j.get('parity_sample'), j['index']['hbm_bytes'])
PY
done
bash tools/kstats.sh r04g --fm9 $FM9 --no-extra-configs --steps 20 | head -10
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --genome repeats --no-extra-configs --no-extras --steps 5 --cpu-seconds 3 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('repeats', round(j['value']/1e6,1), j['ms_per_step'], j['phases_ms'], j['parity_sample'])"
rm -f /dev/shm/dicey_bench_*
