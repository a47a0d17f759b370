#!/bin/bash
# r04 call 17: the lanes' common timeline (dg_hunt_result::t_search_*) and the roofline's busy time per launch: tests, driver-shaped run.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -q -k "submit or device or rccl or ranks" > gpurun_out/r04/pytest_gpu17.log 2>&1
tail -3 gpurun_out/r04/pytest_gpu17.log
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r04/bench_final5.json 2> gpurun_out/r04/bench_final5.err
echo full bench took $SECONDS s; tail -2 gpurun_out/r04/bench_final5.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_final5.json') if l.startswith('{')][-1])
r=j['roofline']
print('value', j['value'], j['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], 'launch_ms', r['launch_ms'], r['busy'], r.get('one_in_flight'), r['request_rate'], j['parity_sample'])
for k in ('host_to_host_pipelined','value_with_d2h','value_same_batch','value_one_in_flight','cli_end_to_end_10M'):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('note','index_open_phases_ms')}
    print(k, v)
for k in j:
    if k.startswith('summary_'): print(k, j[k])
d2=j['extra_configs']['hunt_d2']['roofline']; print('d2', d2['frac'], d2['kernel_ms'], d2['launch_ms'], d2['busy'], d2['traffic'])
PY
