#!/bin/bash
# r03 call 19: locate job kernels only when the previous batch queued jobs: whole suite, default line + kernel stats, repeats line
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03s
rm -rf $OUT; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 900 python bench.py --keep-index --no-extra-configs --cpu-seconds 3 --no-extras > $OUT/bench.json 2> $OUT/bench.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
bash tools/kstats.sh r03s --fm9 $FM9 --parity-queries 0 --no-extra-configs
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --cpu-seconds 3 --no-extras --no-extra-configs --parity-queries 300 > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03s/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"))
PY
