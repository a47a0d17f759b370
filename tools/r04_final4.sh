#!/bin/bash
# r04 final, after the lanes of a handle share what they learn: whole GPU suite, driver-shaped run.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_gpu_final4.log 2>&1
tail -3 gpurun_out/r04/pytest_gpu_final4.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r04/bench_final4.json 2> gpurun_out/r04/bench_final4.err
echo full bench took $SECONDS s; tail -2 gpurun_out/r04/bench_final4.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_final4.json') if l.startswith('{')][-1])
print('value', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms'], j['roofline'].get('one_in_flight'), j['roofline']['traffic'], j['phases_ms'], j['parity_sample'])
for k in ('cli_end_to_end','cli_end_to_end_1M','cli_end_to_end_10M','host_to_host_pipelined','value_with_d2h','value_same_batch','value_one_in_flight'):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('note','index_open_phases_ms')}
    print(k, v)
for k in j:
    if k.startswith('summary_'): print(k, j[k])
for n in ('hunt_d2','hunt_d1_repeats'):
    e=j['extra_configs'][n]; print(n, e['ms_per_step'], e['phases_ms'])
PY
