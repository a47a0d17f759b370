#!/bin/bash
# r04 final with three lanes per handle: whole GPU suite, driver-shaped run.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_gpu_final7.log 2>&1
tail -3 gpurun_out/r04/pytest_gpu_final7.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r04/bench_final7.json 2> gpurun_out/r04/bench_final7.err
echo full bench took $SECONDS s; tail -2 gpurun_out/r04/bench_final7.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_final7.json') if l.startswith('{')][-1])
r=j['roofline']
print('value', j['value'], j['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], 'launch_ms', r['launch_ms'], r['busy'], r.get('one_in_flight'), r['request_rate'], j['parity_sample'])
for k in ('cli_end_to_end','cli_end_to_end_1M','cli_end_to_end_10M','host_to_host_pipelined','value_with_d2h','value_same_batch','value_one_in_flight'):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('note','index_open_phases_ms')}
    print(k, v)
for k in j:
    if k.startswith('summary_'): print(k, j[k])
PY
