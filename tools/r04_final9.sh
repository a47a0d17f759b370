#!/bin/bash
# r04: the driver-shaped run of the final build (k_search2p at five wavefronts per SIMD).
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 290 python bench.py > gpurun_out/r04/bench_final9.json 2> gpurun_out/r04/bench_final9.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_final9.json') if l.startswith('{')][-1])
r=j['roofline']
print('value', j['value'], j['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms'], j['parity_sample'])
for k in ('cli_end_to_end_10M','host_to_host_pipelined'):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('note','index_open_phases_ms')}
    print(k, v)
for k in j:
    if k.startswith('summary_'): print(k, j[k]['value'], j[k]['ms_per_step'], j[k]['parity'])
PY
