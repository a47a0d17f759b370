#!/bin/bash
# r03 call 5: memo verify + compact ops: parity tests, then the repeats / default / d2 bench lines
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03g
rm -rf $OUT; mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_locate_topk.py tests/test_gpu_capped.py tests/test_gpu_cli.py tests/test_gpu_multirank.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 900 python bench.py --keep-index --no-extra-configs --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --no-extra-configs --no-cpu-baseline --no-extras --parity-queries 100 > $OUT/bench_d2.json 2> $OUT/bench_d2.err
rm -f /dev/shm/dicey_bench_*
for ch in 0 4 1; do
DICEY_VERIFY_CH=$ch timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries $([ $ch = 0 ] && echo 300 || echo 0) --keep-index $([ $ch != 0 ] && echo --fm9 $(ls /dev/shm/dicey_bench_*.fm9 | head -1)) > $OUT/bench_repeats_ch$ch.json 2> $OUT/bench_repeats_ch$ch.err
done
rm -f /dev/shm/dicey_bench_*
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03g/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], d["value"], d["ms_per_step"], d["phases_ms"], d.get("parity_sample"), (d.get("value_with_d2h") or {}).get("value"))
PY
