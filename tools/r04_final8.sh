#!/bin/bash
# r04: the driver-shaped run once more (host-to-host pass on three lanes as well).
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r04/bench_final8.json 2> gpurun_out/r04/bench_final8.err
echo full bench took $SECONDS s; tail -2 gpurun_out/r04/bench_final8.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_final8.json') if l.startswith('{')][-1])
r=j['roofline']
print('value', j['value'], j['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], r['busy'], r.get('one_in_flight'), j['parity_sample'])
for k in ('cli_end_to_end_10M','host_to_host_pipelined','value_with_d2h','value_one_in_flight'):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('note','index_open_phases_ms')}
    print(k, v)
for k in j:
    if k.startswith('summary_'): print(k, j[k]['value'], j[k]['ms_per_step'], j[k]['parity'])
PY
