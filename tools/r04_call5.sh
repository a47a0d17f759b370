#!/bin/bash
# r04 call 5: same-box A/B of the r03 and r04 forms of k_search1s; then the driver-shaped run (extras, extra_configs, CLI at 1 M / 10 M)
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python bench.py --no-extra-configs --no-cpu-baseline --no-extras --steps 30 --keep-index > gpurun_out/r04/ab_c.json 2>/dev/null
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
DICEY_NO_PREP_FUSION=1 timeout 600 python bench.py --fm9 $FM9 --no-extra-configs --no-cpu-baseline --no-extras --steps 30 > gpurun_out/r04/ab_b.json 2>/dev/null
cp gpurun_out/r04/ab_b.json gpurun_out/r04/ab_a.json
timeout 600 python bench.py --fm9 $FM9 --no-extra-configs --no-cpu-baseline --no-extras --steps 30 > gpurun_out/r04/ab_c2.json 2>/dev/null
for f in ab_c ab_b ab_a ab_c2; do python - $f <<'PY'
import json,sys
j=json.loads([l for l in open('gpurun_out/r04/%s.json'%sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], round(j['value']/1e6,1), 'M/s', round(j['ms_per_step'],4), {k: round(v,4) for k,v in j['phases_ms'].items()})
PY
done
bash tools/kstats.sh r04e --fm9 $FM9 --no-extra-configs --steps 20
rm -f /dev/shm/dicey_bench_*
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r04/bench_full.json 2> gpurun_out/r04/bench_full.err
echo full bench took $SECONDS s; tail -3 gpurun_out/r04/bench_full.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_full.json') if l.startswith('{')][-1])
print('value', j['value'], j['ms_per_step'])
for k in ('cli_end_to_end','cli_end_to_end_1M','cli_end_to_end_10M','host_to_host_pipelined','value_with_d2h'):
    v=j.get(k); 
    if isinstance(v,dict): v={a:b for a,b in v.items() if a not in ('note','index_open_phases_ms')}
    print(k, v)
for k,v in (j.get('extra_configs') or {}).items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms_per_step'), v.get('parity_sample'), v.get('error'))
    else: print(k, v)
PY
