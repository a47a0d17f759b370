#!/bin/bash
# randomised differential runs of dg_hunt against the checker in the FULL-SIZE layout (table order 17, long filter 18: the flat distance-2
# kernel's LONG2 body, the r05 rebuild) — tools/fuzz_hunt.py seeds $SEEDS, 40 configurations each
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
export DICEY_KMER_K=17 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1
for S in $SEEDS; do
  timeout 900 python tools/fuzz_hunt.py $S 40 > gpurun_out/r05/fuzz_full_$S.log 2>&1
  echo "seed $S: $(grep -c ' ok$' gpurun_out/r05/fuzz_full_$S.log) ok, $(tail -1 gpurun_out/r05/fuzz_full_$S.log)"
done
grep -h "MISMATCH\|refused" gpurun_out/r05/fuzz_full_*.log | head -10
