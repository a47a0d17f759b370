#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03b
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --no-cpu-baseline --no-extras --parity-queries 300 > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03b/bench_repeats.json"))
print("repeats:", d["value"], d["ms_per_step"], d["phases_ms"], d["hits_per_step"], d.get("parity_sample"))
PY
