#!/bin/bash
# GPU box experiment: the default bench (and optionally hunt_d2) under a list of environment settings.
# usage: tools/exp_env.sh "<cfgs>" "VAR=1" "VAR=2 OTHER=3" ...
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/expe; mkdir -p $OUT
FM9=$(ls /dev/shm/dicey_bench_*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --parity-queries 0 --keep-index > $OUT/build.json 2> $OUT/build.err
  FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
fi
CFGS=$1; shift
i=0
for E in "$@"; do
  i=$((i+1))
  for CFG in $CFGS; do
    env $E timeout 300 python bench.py --fm9 $FM9 --config $CFG --steps 10 --warmup 2 --no-extras --no-cpu-baseline --parity-queries 0 > $OUT/e${i}_$CFG.json 2> $OUT/e${i}_$CFG.err
    python - <<PY
import json
try:
    j = json.load(open("$OUT/e${i}_$CFG.json")); r = j["roofline"]
    print("$CFG [$E]", "value", round(j["value"]), "ms/step", round(j["ms_per_step"], 4), {k: round(v, 4) for k, v in j["phases_ms"].items()})
except Exception as e:
    print("$CFG [$E] failed", e)
PY
  done
done
