#!/bin/bash
# r04 call 19: probe loop of k_search2p as a software pipeline over the first operation (two rounds of loads in flight):
# base build / pipelined at 3 wavefronts per SIMD (155 VGPRs) / pipelined at 4 (128 VGPRs, 64 B scratch), same box, same index.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
cp dicey_amd/libdiceygpu_k2pbase.so dicey_amd/libdiceygpu.so
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
for v in base A B base A B; do
  cp dicey_amd/libdiceygpu_k2p$v.so dicey_amd/libdiceygpu.so
  timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 6 --warmup 3 --cpu-seconds 3 --parity-queries 300 --no-extras --no-extra-configs --in-flight 1 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 $v', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'], j['parity_sample'])
open('gpurun_out/r04/d2_pipe_$v.json','w').write(json.dumps(j))"
done
for v in A B; do
  cp dicey_amd/libdiceygpu_k2p$v.so dicey_amd/libdiceygpu.so
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "mode or distance_2 or d2 or fuzz" > gpurun_out/r04/pytest_gpu19_$v.log 2>&1
  tail -2 gpurun_out/r04/pytest_gpu19_$v.log
  FUZZ_FAST_NEIGHBORS=1 timeout 300 python tools/fuzz_hunt.py 71 60 2>&1 | tail -1
done
rm -f /dev/shm/dicey_bench_*
