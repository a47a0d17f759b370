#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03e
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py tests/test_gpu_capped.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
bash tools/kstats.sh r03e_repeats --genome repeats --steps 5 --warmup 2 --parity-queries 0
python - <<'PY'
import json
d = json.load(open("gpurun_out/kstats_r03e_repeats/bench.json"))
print("repeats (traced):", d["value"], d["ms_per_step"], d["phases_ms"], d["hits_per_step"])
PY
