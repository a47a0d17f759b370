#!/bin/bash
# r03 call 17: locate job kernels side by side on helper streams: locate / parity / multirank modules, repeats line with parity, default line
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03r
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --cpu-seconds 3 --no-extras --no-extra-configs --parity-queries 300 --keep-index > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 > $OUT/bench_repeats_20.json 2> $OUT/bench_repeats_20.err
bash tools/kstats.sh r03r_repeats --genome repeats --fm9 $FM9 --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --no-extra-configs --no-cpu-baseline --no-extras --parity-queries 300 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03r/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"))
PY
