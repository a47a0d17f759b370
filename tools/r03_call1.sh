#!/bin/bash
# r03 call 1: repeat-genome job statistics (DICEY_DUMP_JOBS) + the gather ceiling matrix (sizes x lanes x dependent / independent, filter geometry)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03a
rm -rf $OUT; mkdir -p $OUT
G=tools/microbench/gather_bench
for gib in 4 16 64 170; do
  timeout 300 $G $gib 64 2 200000,2000000,6000000 >> $OUT/gather_matrix.jsonl 2>> $OUT/gather.err
done
for cg in 2 8.6; do
  for lps in 6 12 24; do
    timeout 300 $G filter $cg 200000 $lps >> $OUT/gather_filter.jsonl 2>> $OUT/gather.err
  done
done
DICEY_DUMP_JOBS=$GRAFT_REPO_ROOT/$OUT/jobs.bin timeout 900 python bench.py --genome repeats --steps 3 --warmup 1 --no-cpu-baseline --no-extras --parity-queries 0 > $OUT/bench_repeats.json 2> $OUT/bench_repeats.err
python - <<'PY'
import struct, numpy as np, json
d = open("gpurun_out/r03a/jobs.bin", "rb").read()
n = struct.unpack("<I", d[:4])[0]
a = np.frombuffer(d[4:4 + 12 * n], dtype=np.uint32).reshape(-1, 3)
occs, take = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
edges = [24, 64, 256, 1024, 4096, 8192, 32768, 131072, 1 << 20, 1 << 32]
rows = []
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (occs > lo) & (occs <= hi)
    rows.append({"occs": f"({lo},{hi}]", "jobs": int(m.sum()), "sum_occs": int(occs[m].sum()), "sum_take": int(take[m].sum()),
                 "take_le_1024": int((m & (take <= 1024)).sum())})
json.dump({"jobs": int(n), "rows": rows, "take_hist": np.bincount(np.minimum(take, 1024) // 128).tolist()}, open("gpurun_out/r03a/job_stats.json", "w"), indent=1)
print(open("gpurun_out/r03a/job_stats.json").read())
PY
cat $OUT/gather_matrix.jsonl | tail -80
cat $OUT/gather_filter.jsonl
