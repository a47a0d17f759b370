#!/bin/bash
# r04 call 13: select stage inside k_search2p (distance 2), dg_hunt_device_submit (two batches in flight): parity, fuzz, A/Bs.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_layout.py tests/test_gpu_capped.py tests/test_gpu_multirank.py -x -q > gpurun_out/r04/pytest_gpu13.log 2>&1
tail -4 gpurun_out/r04/pytest_gpu13.log
for sd in 51 52 53 54; do FUZZ_FAST_NEIGHBORS=1 timeout 300 python tools/fuzz_hunt.py $sd 60 2>&1 | tail -1; done
for sd in 55 56; do FUZZ_FAST_NEIGHBORS=1 DICEY_KMER_K=10 DICEY_KMER_K2=14 timeout 300 python tools/fuzz_hunt.py $sd 60 2>&1 | tail -1; done
for sd in 57; do FUZZ_FAST_NEIGHBORS=1 DICEY_FUSED_LCAP=6 timeout 300 python tools/fuzz_hunt.py $sd 60 2>&1 | tail -1; done
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
for v in fused generic fused generic; do
  if [ $v = generic ]; then export DICEY_NO_FUSED_SELECT2=1; else unset DICEY_NO_FUSED_SELECT2; fi
  timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 6 --warmup 2 --cpu-seconds 3 --parity-queries 300 --no-extras --no-extra-configs --in-flight 1 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 $v', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'], j['parity_sample'], j['roofline']['filter_probes_per_launch'], j['roofline']['ext_steps_per_launch'])
open('gpurun_out/r04/d2_sel_$v.json','w').write(json.dumps(j))"
done
unset DICEY_NO_FUSED_SELECT2
timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 6 --warmup 2 --cpu-seconds 3 --parity-queries 0 --no-extras --no-extra-configs --in-flight 2 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 two in flight', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'])"
for v in 2 1 2 1; do
  timeout 600 python bench.py --fm9 $FM9 --steps 30 --warmup 3 --no-cpu-baseline --parity-queries 1000 --no-extras --no-extra-configs --in-flight $v 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d1 in flight $v', round(j['value']/1e6,2), j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['phases_ms'], j['parity_sample'])
open('gpurun_out/r04/d1_inflight_$v.json','w').write(json.dumps(j))"
done
DICEY_TIMING=2 timeout 600 python bench.py --fm9 $FM9 --steps 6 --warmup 2 --no-cpu-baseline --parity-queries 0 --no-extras --no-extra-configs --in-flight 2 2>&1 | grep "dicey timing" | tail -24
rm -f /dev/shm/dicey_bench_*
