#!/bin/bash
# Experiment (GPU box): table order / long-filter order variants on the bench workloads.  usage: tools/exp_k.sh "K:K2 K:K2 ..."
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/expk
mkdir -p $OUT
FM9=$(ls /dev/shm/dicey_bench_*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --parity-queries 0 --keep-index > $OUT/build.json 2> $OUT/build.err
  FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
fi
for V in ${1:-"17:18 16:18"}; do
  K=${V%%:*}; K2=${V##*:}
  for CFG in hunt_d1 hunt_d2; do
    DICEY_KMER_K=$K DICEY_KMER_K2=$K2 timeout 300 python bench.py --fm9 $FM9 --config $CFG --steps 5 --warmup 2 --no-extras --no-cpu-baseline --parity-queries 0 \
      > $OUT/${CFG}_K${K}_K2${K2}.json 2> $OUT/${CFG}_K${K}_K2${K2}.err
    python - <<PY
import json
try:
    j = json.load(open("$OUT/${CFG}_K${K}_K2${K2}.json"))
    r = j["roofline"]
    print("$CFG K=$K K2=$K2", "value", round(j["value"]), "ms/step", round(j["ms_per_step"], 3), "kernel_ms", round(r["kernel_ms"], 3), "ext", r["ext_steps_per_launch"], "tab", r["table_reads_per_launch"], "probes", r["filter_probes_per_launch"], "hbm_GB", round(j["index"]["hbm_bytes"] / 1e9, 1))
except Exception as e:
    print("$CFG K=$K K2=$K2 failed", e)
PY
  done
done
