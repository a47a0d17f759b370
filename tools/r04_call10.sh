#!/bin/bash
# r04 call 10: k_search2p with the filter copy picked by the second operation's position (A/B + parity), then the driver-shaped run.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
for v in new old; do
  if [ $v = old ]; then export DICEY_EXP_COPY_P1=1; else unset DICEY_EXP_COPY_P1; fi
  timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 4 --warmup 1 --cpu-seconds 3 --parity-queries 300 --no-extras --no-extra-configs 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 $v', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'], j['parity_sample'], j['roofline']['filter_probes_per_launch'], j['roofline']['ext_steps_per_launch'])"
done
unset DICEY_EXP_COPY_P1
bash tools/prof_cfg.sh r04d2b --config hunt_d2 --no-extra-configs | grep -E "k_search2p"
rm -f /dev/shm/dicey_bench_*
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_layout.py -x -q -k "distance or d2 or edit2 or two or layout or mode" > gpurun_out/r04/pytest_gpu7.log 2>&1
tail -3 gpurun_out/r04/pytest_gpu7.log
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err
echo full bench took $SECONDS s; tail -2 gpurun_out/r04/bench_final.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_final.json') if l.startswith('{')][-1])
print('value', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'])
for k in ('cli_end_to_end','cli_end_to_end_1M','cli_end_to_end_10M','host_to_host_pipelined','value_with_d2h','value_same_batch'):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('note','index_open_phases_ms')}
    print(k, v)
for k in j:
    if k.startswith('summary_'): print(k, j[k])
PY
