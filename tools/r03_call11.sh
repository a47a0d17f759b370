#!/bin/bash
# r03 call 11: select by the first wavefront; context characters from the index (sai + k_sel_bounds): tests, default, repeats with / without bounds
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03m
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py tests/test_gpu_capped.py tests/test_gpu_fullsize_layout.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
timeout 900 python bench.py --no-extra-configs --no-cpu-baseline --no-extras --parity-queries 0 > $OUT/bench.json 2> $OUT/bench.err
timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --cpu-seconds 3 --no-extras --no-extra-configs --parity-queries 300 --keep-index > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
DICEY_NO_CTX_BOUNDS=1 timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 > $OUT/bench_repeats_nobounds.json 2> $OUT/bench_repeats_nobounds.err
DICEY_DBG_VERIFY=4 timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 > $OUT/bench_repeats_check.json 2> $OUT/bench_repeats_check.err
bash tools/kstats.sh r03m_repeats --genome repeats --fm9 $FM9 --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs
rm -f /dev/shm/dicey_bench_*
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03m/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"))
PY
tail -3 $OUT/bench_repeats_check.err
