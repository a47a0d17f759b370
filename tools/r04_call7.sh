#!/bin/bash
# r04 call 7: k_cap_enum with probes / inserts in flight together (tests + 25-mer line + kernel stats), locate-job duplicates on the repeats
# genome, the default step's kernel stats after the verify change.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_capped.py -x -q > gpurun_out/r04/pytest_gpu5.log 2>&1
tail -3 gpurun_out/r04/pytest_gpu5.log
timeout 600 python bench.py --steps 20 --no-extras --no-cpu-baseline --no-extra-configs --keep-index 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default', round(j['value']/1e6,1), j['ms_per_step'], j['phases_ms'])"
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 900 python bench.py --fm9 $FM9 --config hunt_d2 --qlen 25 --queries 2000 --steps 3 --warmup 1 --cpu-seconds 4 --parity-queries 100 --no-extras --no-extra-configs > gpurun_out/r04/bench_25b.json 2> gpurun_out/r04/bench_25b.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_25b.json') if l.startswith('{')][-1])
print('25mers', j['value'], j['ms_per_step'], j['phases_ms'], j.get('cap_stage'), j.get('parity_sample'), j['roofline'].get('kernel'), j['roofline'].get('frac'))
PY
bash tools/kstats.sh r04_25b --fm9 $FM9 --config hunt_d2 --qlen 25 --queries 2000 --steps 3 --warmup 1 --no-extra-configs | grep -E "cap_enum|explicit|search2p|group_select"
bash tools/kstats.sh r04f --fm9 $FM9 --no-extra-configs --steps 20 | head -12
rm -f /dev/shm/dicey_bench_*
DICEY_DUMP_JOBS=$GRAFT_REPO_ROOT/gpurun_out/r04/jobs.bin timeout 900 python bench.py --genome repeats --no-extra-configs --no-extras --no-cpu-baseline --steps 3 --warmup 1 --batches 1 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('repeats', round(j['value']/1e6,1), j['ms_per_step'], j['phases_ms'], j['hits_per_step'])"
python - <<'PY'
import numpy as np
a=np.fromfile('gpurun_out/r04/jobs.bin',dtype=np.uint32).reshape(-1,4)
print('jobs',len(a),'sum take',int(a[:,2].sum()))
key=a[:,0].astype(np.uint64)<<32|a[:,1]
u,idx,cnt=np.unique(key,return_index=True,return_counts=True)
print('unique (lo,occs)',len(u),'sum take of unique',int(a[idx,2].sum()))
big=a[:,1]>256
kb=key[big]; ub,ib=np.unique(kb,return_index=True)
print('big jobs',int(big.sum()),'unique big',len(ub),'take big',int(a[big,2].sum()),'take unique big',int(a[big][ib,2].sum()))
PY
rm -f gpurun_out/r04/jobs.bin /dev/shm/dicey_bench_*
