#!/bin/bash
# r03 call 7: tail changes (k_prepare words, batch_finish in the last verify workgroup, chained scans) through the whole GPU suite,
# then the default line, d2, and a cost split of k_verify_memo on the repeats genome
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03i
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
timeout 900 python bench.py --keep-index --no-extra-configs --cpu-seconds 4 > $OUT/bench.json 2> $OUT/bench.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 600 python bench.py --fm9 $FM9 --big-table --no-extra-configs --no-cpu-baseline --no-extras --parity-queries 0 > $OUT/bench_bigtable.json 2> $OUT/bench_bigtable.err
timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --no-extra-configs --no-cpu-baseline --no-extras --parity-queries 0 > $OUT/bench_d2.json 2> $OUT/bench_d2.err
bash tools/kstats.sh r03i --fm9 $FM9 --parity-queries 0 --no-extra-configs
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --genome repeats --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 --keep-index > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
for dbg in 1 2 3; do
DICEY_DBG_VERIFY=$dbg timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 > $OUT/bench_repeats_dbg$dbg.json 2> $OUT/bench_repeats_dbg$dbg.err
done
rm -f /dev/shm/dicey_bench_*
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03i/bench*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"), round((d.get("value_with_d2h") or {}).get("value", 0)))
PY
