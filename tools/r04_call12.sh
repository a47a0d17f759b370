#!/bin/bash
# r04 call 12: filtered intervals through the distance-2 path (k_search2p -> k_group_pack -> Sel): parity subset, fuzz, A/B, profile.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_layout.py tests/test_gpu_capped.py -x -q > gpurun_out/r04/pytest_gpu12.log 2>&1
tail -4 gpurun_out/r04/pytest_gpu12.log
for sd in 41 42 43 44; do FUZZ_FAST_NEIGHBORS=1 timeout 300 python tools/fuzz_hunt.py $sd 60 2>&1 | tail -1; done
for sd in 45 46; do FUZZ_FAST_NEIGHBORS=1 DICEY_KMER_K=10 DICEY_KMER_K2=14 timeout 300 python tools/fuzz_hunt.py $sd 60 2>&1 | tail -1; done
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
for v in new old new old; do
  if [ $v = old ]; then export DICEY_NO_PRE5_D2=1; else unset DICEY_NO_PRE5_D2; fi
  timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 4 --warmup 1 --cpu-seconds 3 --parity-queries 300 --no-extras --no-extra-configs 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 $v', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'], j['parity_sample'], j['roofline']['filter_probes_per_launch'], j['roofline']['ext_steps_per_launch'])
open('gpurun_out/r04/d2_pre5_$v.json','w').write(json.dumps(j))"
done
unset DICEY_NO_PRE5_D2
bash tools/prof_cfg.sh r04d2c --config hunt_d2 --no-extra-configs | grep -E "k_search2p|k_group|k_leaf|k_verify|k_locate"
rm -f /dev/shm/dicey_bench_*
