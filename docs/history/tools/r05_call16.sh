#!/bin/bash
# randomised differential runs of the binary's `search` and `padlock` against the checker (reference thal.h): thal wave kernels after the r05
# changes (END1 from the end tables, k_thal_self_wave<INSIDE> for arm lengths 15-28), and more hunt seeds in the full-size layout
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for S in $PSEEDS; do
  timeout 900 python tools/fuzz_padlock.py $S 12 > gpurun_out/r05/fuzz_padlock_$S.log 2>&1; echo "padlock seed $S: $(tail -1 gpurun_out/r05/fuzz_padlock_$S.log)"
done
for S in $SSEEDS; do
  timeout 900 python tools/fuzz_search.py $S 12 > gpurun_out/r05/fuzz_search_$S.log 2>&1; echo "search seed $S: $(tail -1 gpurun_out/r05/fuzz_search_$S.log)"
done
export DICEY_KMER_K=17 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1
for S in $HSEEDS; do
  timeout 900 python tools/fuzz_hunt.py $S 40 > gpurun_out/r05/fuzz_full_$S.log 2>&1
  echo "hunt seed $S: $(grep -c ' ok$' gpurun_out/r05/fuzz_full_$S.log) ok, $(tail -1 gpurun_out/r05/fuzz_full_$S.log)"
done
grep -h "MISMATCH\|DIFF\|differ" gpurun_out/r05/fuzz_*.log | head -10
