#!/bin/bash
# r03 call 14: k_search2p<true> (select inside the distance-2 search kernel): fused test, d2 modules, d2 bench with parity, A/B without
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03p
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_capped.py tests/test_gpu_fullsize_layout.py tests/test_gpu_padlock.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 900 python bench.py --config hunt_d2 --cpu-seconds 3 --no-extra-configs --no-extras --parity-queries 300 --keep-index > $OUT/bench_d2.json 2> $OUT/bench_d2.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
DICEY_NO_FUSED_SELECT2=1 timeout 600 python bench.py --config hunt_d2 --fm9 $FM9 --no-cpu-baseline --no-extra-configs --no-extras --parity-queries 0 > $OUT/bench_d2_unfused.json 2> $OUT/bench_d2_unfused.err
bash tools/kstats.sh r03p_d2 --config hunt_d2 --fm9 $FM9 --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs
rm -f /dev/shm/dicey_bench_*
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03p/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"))
PY
