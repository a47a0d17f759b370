#!/bin/bash
# r03 call 16: two buffer sizes in k_locate_topk: locate / parity modules, repeats line with parity, A/B with one size
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03q
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --cpu-seconds 3 --no-extras --no-extra-configs --parity-queries 300 --keep-index > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
DICEY_TOPK_ONE_SIZE=1 timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 > $OUT/bench_repeats_onesize.json 2> $OUT/bench_repeats_onesize.err
DICEY_VERIFY_CH=4 timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 > $OUT/bench_repeats_ch4.json 2> $OUT/bench_repeats_ch4.err
timeout 600 python bench.py --genome repeats --fm9 $FM9 --big-table --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 > $OUT/bench_repeats_bigtable.json 2> $OUT/bench_repeats_bigtable.err
bash tools/kstats.sh r03q_repeats --genome repeats --fm9 $FM9 --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs
rm -f /dev/shm/dicey_bench_*
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03q/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"))
PY
