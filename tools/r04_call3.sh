#!/bin/bash
# r04 call 3: A/B of the early wavefront exit in k_search1s, host timing of the submit / wait pipeline, GPU tests of the capped
# device path and the rest of the suite behind the point call 2 stopped at.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python bench.py --no-extra-configs --no-cpu-baseline --no-extras --steps 20 --keep-index > gpurun_out/r04/ab_leave.json 2> gpurun_out/r04/ab_leave.err
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
DICEY_EXP_NOLEAVE=1 timeout 600 python bench.py --fm9 $FM9 --no-extra-configs --no-cpu-baseline --no-extras --steps 20 > gpurun_out/r04/ab_noleave.json 2> gpurun_out/r04/ab_noleave.err
for f in ab_leave ab_noleave; do python - $f <<'PY'
import json,sys
j=json.loads([l for l in open('gpurun_out/r04/%s.json'%sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], round(j['value']/1e6,1), 'M/s', j['phases_ms'])
PY
done
DICEY_TIMING=2 timeout 600 python bench.py --fm9 $FM9 --no-extra-configs --no-cpu-baseline --steps 10 > gpurun_out/r04/bench_c.json 2> gpurun_out/r04/bench_c.err
grep "dicey timing" gpurun_out/r04/bench_c.err | grep -E "submit|wait|worker|batch of 100000" | tail -60 > gpurun_out/r04/timing_tail.txt
tail -40 gpurun_out/r04/timing_tail.txt
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04/bench_c.json') if l.startswith('{')][-1])
print('h2h', j.get('host_to_host_pipelined')); print('d2h', j.get('value_with_d2h')); print('value', j['value'], j['phases_ms'])
PY
rm -f /dev/shm/dicey_bench_*
timeout 1500 python -m pytest tests/test_gpu_capped.py tests/test_gpu_multirank.py tests/test_gpu_padlock.py tests/test_gpu_parity.py tests/test_gpu_search.py tests/test_gpu_thal_wave.py -x -q > gpurun_out/r04/pytest_gpu2.log 2>&1
tail -8 gpurun_out/r04/pytest_gpu2.log
