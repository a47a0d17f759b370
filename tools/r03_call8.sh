#!/bin/bash
# r03 call 8: after the revert of the in-kernel fusions: parity modules incl. submit/wait and the CLI pipeline, default line + kernel stats, repeats
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03j
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_capped.py tests/test_gpu_locate_topk.py tests/test_gpu_multirank.py tests/test_gpu_search.py tests/test_gpu_padlock.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
timeout 900 python bench.py --keep-index --no-extra-configs --cpu-seconds 4 > $OUT/bench.json 2> $OUT/bench.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --no-extra-configs --no-cpu-baseline --no-extras --parity-queries 0 > $OUT/bench_d2.json 2> $OUT/bench_d2.err
bash tools/kstats.sh r03j --fm9 $FM9 --parity-queries 0 --no-extra-configs
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03j/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"), round((d.get("value_with_d2h") or {}).get("value", 0)), d.get("cli_end_to_end"), d.get("cli_end_to_end_after_release"))
PY
