#!/bin/bash
# r03 final (second pass, final build): profile round of the default configuration, kernel stats of repeats / d2, then the
# driver-shaped default run
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r03zz
rm -rf $OUT; mkdir -p $OUT
bash tools/profile_round.sh r03 > $OUT/profile_round.log 2>&1
tail -3 $OUT/profile_round.log
timeout 900 python bench.py --keep-index --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
bash tools/kstats.sh r03zz_d2 --config hunt_d2 --fm9 $FM9 --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs > $OUT/kstats_d2.log 2>&1
rm -f /dev/shm/dicey_bench_*
bash tools/kstats.sh r03zz_repeats --genome repeats --steps 5 --warmup 2 --parity-queries 0 --no-extra-configs > $OUT/kstats_repeats.log 2>&1
rm -f /dev/shm/dicey_bench_*
( time timeout 1200 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r03zz/bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("default:", round(d["value"]), d["ms_per_step"], {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"))
        print(" d2h:", (d.get("value_with_d2h") or {}).get("value"), " h2h:", (d.get("host_to_host_pipelined") or {}).get("value"), " cli:", (d.get("cli_end_to_end") or {}).get("value"), (d.get("cli_end_to_end_after_release") or {}).get("value"))
        for k, v in ((kk, vv) for kk, vv in (d.get("extra_configs") or {}).items() if isinstance(vv, dict)):
            print(k, {a: v.get(a) for a in ("value", "unit", "ms_per_step", "parity_sample", "error")})
PY
