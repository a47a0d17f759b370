#!/bin/bash
# r04 call 14: select stage of k_search2p shared by all lanes, one call site of the search: parity, fuzz, A/B against the generic select kernels.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capped.py tests/test_gpu_multirank.py -x -q > gpurun_out/r04/pytest_gpu14.log 2>&1
tail -4 gpurun_out/r04/pytest_gpu14.log
for sd in 61 62 63; do FUZZ_FAST_NEIGHBORS=1 timeout 300 python tools/fuzz_hunt.py $sd 60 2>&1 | tail -1; done
for sd in 65; do FUZZ_FAST_NEIGHBORS=1 DICEY_KMER_K=10 DICEY_KMER_K2=14 timeout 300 python tools/fuzz_hunt.py $sd 60 2>&1 | tail -1; done
for sd in 67; do FUZZ_FAST_NEIGHBORS=1 DICEY_FUSED_LCAP=6 timeout 300 python tools/fuzz_hunt.py $sd 60 2>&1 | tail -1; done
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
for v in fused generic noprep fused generic; do
  unset DICEY_NO_FUSED_SELECT2 DICEY_NO_PREP_FUSION
  if [ $v = generic ]; then export DICEY_NO_FUSED_SELECT2=1; fi
  if [ $v = noprep ]; then export DICEY_NO_PREP_FUSION=1; fi
  timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 6 --warmup 2 --cpu-seconds 3 --parity-queries 300 --no-extras --no-extra-configs --in-flight 1 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 $v', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'], j['parity_sample'])
open('gpurun_out/r04/d2_sel2_$v.json','w').write(json.dumps(j))"
done
unset DICEY_NO_FUSED_SELECT2 DICEY_NO_PREP_FUSION
timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 8 --warmup 2 --cpu-seconds 3 --parity-queries 0 --no-extras --no-extra-configs --in-flight 2 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 two in flight', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'])"
rm -f /dev/shm/dicey_bench_*
