set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03fuzz2
for s in 41 42 43 44; do FUZZ_FAST_NEIGHBORS=1 timeout 900 python tools/fuzz_hunt.py $s 40 > gpurun_out/r03fuzz2/hunt_$s.log 2>&1; tail -1 gpurun_out/r03fuzz2/hunt_$s.log; done
timeout 900 python tools/fuzz_hunt.py 45 30 > gpurun_out/r03fuzz2/hunt_literal_45.log 2>&1; tail -1 gpurun_out/r03fuzz2/hunt_literal_45.log
for s in 41 42; do timeout 600 python tools/fuzz_search.py $s 25 > gpurun_out/r03fuzz2/search_$s.log 2>&1; tail -1 gpurun_out/r03fuzz2/search_$s.log; done
for s in 41 42; do timeout 600 python tools/fuzz_padlock.py $s 25 > gpurun_out/r03fuzz2/padlock_$s.log 2>&1; tail -1 gpurun_out/r03fuzz2/padlock_$s.log; done
grep -h "refused\|MISMATCH" gpurun_out/r03fuzz2/*.log | head
