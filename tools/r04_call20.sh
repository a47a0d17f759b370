#!/bin/bash
# r04 call 20: a third lane per handle: two against three batches in flight on one box (distance 1, repeats genome, distance 2), lane tests.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py tests/test_gpu_cli.py -q -k "submit or device or rccl or ranks or pipeline" > gpurun_out/r04/pytest_gpu20.log 2>&1
tail -3 gpurun_out/r04/pytest_gpu20.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
for v in 3 2 3 2 1; do
  timeout 600 python bench.py --fm9 $FM9 --steps 30 --warmup 6 --no-cpu-baseline --parity-queries 1000 --no-extras --no-extra-configs --in-flight $v 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']; print('d1 in flight $v', round(j['value']/1e6,2), j['ms_per_step'], r['kernel_ms'], r['launch_ms'], r['frac'], r['busy'], j['parity_sample'])
open('gpurun_out/r04/d1_lanes_$v.json','w').write(json.dumps(j))"
done
for v in 3 2; do
  timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 9 --warmup 6 --cpu-seconds 3 --parity-queries 0 --no-extras --no-extra-configs --in-flight $v 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 in flight $v', round(j['value']/1e6,2), j['ms_per_step'])"
done
rm -f /dev/shm/dicey_bench_*
for v in 3 2; do
  timeout 900 python bench.py --config hunt_d1 --genome repeats --steps 9 --warmup 6 --no-cpu-baseline --parity-queries 300 --no-extras --no-extra-configs --in-flight $v $( [ $v = 3 ] && echo --keep-index || echo --fm9 $(ls /dev/shm/dicey_bench_*rep*.fm9 2>/dev/null | head -1) ) 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('repeats in flight $v', round(j['value']/1e6,2), j['ms_per_step'], j['parity_sample'])"
done
rm -f /dev/shm/dicey_bench_*
