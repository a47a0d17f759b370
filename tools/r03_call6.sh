#!/bin/bash
# r03 call 6: locate in registers / sorted by shuffles: parity, repeats line, d2 line with the new verify heuristic
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03h
rm -rf $OUT; mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_locate_topk.py tests/test_gpu_capped.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --cpu-seconds 4 --no-extras --no-extra-configs --parity-queries 300 > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
timeout 900 python bench.py --config hunt_d2 --cpu-seconds 4 --no-extra-configs --no-extras --parity-queries 200 > $OUT/bench_d2.json 2> $OUT/bench_d2.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03h/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], d["value"], d["ms_per_step"], d["phases_ms"], d.get("parity_sample"))
PY
bash tools/kstats.sh r03h_repeats --genome repeats --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs
