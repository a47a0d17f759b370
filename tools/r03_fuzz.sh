set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03fuzz
for s in 31 32 33; do FUZZ_FAST_NEIGHBORS=1 timeout 900 python tools/fuzz_hunt.py $s 40 > gpurun_out/r03fuzz/hunt_$s.log 2>&1; tail -1 gpurun_out/r03fuzz/hunt_$s.log; done
for s in 31 32; do timeout 600 python tools/fuzz_search.py $s 25 > gpurun_out/r03fuzz/search_$s.log 2>&1; tail -1 gpurun_out/r03fuzz/search_$s.log; done
for s in 31 32; do timeout 600 python tools/fuzz_padlock.py $s 25 > gpurun_out/r03fuzz/padlock_$s.log 2>&1; tail -1 gpurun_out/r03fuzz/padlock_$s.log; done
grep -h "refused\|MISMATCH" gpurun_out/r03fuzz/*.log | head
