#!/bin/bash
# r04 call 22: k_search2p compiled for five wavefronts per SIMD (96 VGPRs, no scratch) against the committed four (109), same box.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
cp dicey_amd/libdiceygpu_k2pbase.so dicey_amd/libdiceygpu.so
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
  FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
fi
for v in base occ5 base occ5; do
  cp dicey_amd/libdiceygpu_k2p$v.so dicey_amd/libdiceygpu.so
  timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 6 --warmup 3 --cpu-seconds 3 --parity-queries 300 --no-extras --no-extra-configs --in-flight 1 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 $v', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms']['ms_search_flat'], j['parity_sample'])
open('gpurun_out/r04/d2_occ_$v.json','w').write(json.dumps(j))"
done
cp dicey_amd/libdiceygpu_k2pocc5.so dicey_amd/libdiceygpu.so
timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 9 --warmup 6 --cpu-seconds 3 --parity-queries 0 --no-extras --no-extra-configs 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 occ5 three in flight', round(j['value']/1e6,2), j['ms_per_step'])"
