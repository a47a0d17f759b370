#!/bin/bash
# r04 final (after the filtered intervals): whole GPU suite, profile round r04b, driver-shaped run.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_final2.log 2>&1
tail -5 gpurun_out/r04/pytest_gpu_final2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
bash tools/profile_round.sh r04b > gpurun_out/r04/profile_round_b.log 2>&1
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r04/bench_final2.json 2> gpurun_out/r04/bench_final2.err
echo full bench took $SECONDS s; tail -2 gpurun_out/r04/bench_final2.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_final2.json') if l.startswith('{')][-1])
print('value', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['phases_ms'], j['parity_sample'])
for k in ('cli_end_to_end','cli_end_to_end_1M','cli_end_to_end_10M','host_to_host_pipelined','value_with_d2h','value_same_batch'):
    v=j.get(k)
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('note','index_open_phases_ms')}
    print(k, v)
for k in j:
    if k.startswith('summary_'): print(k, j[k])
PY
