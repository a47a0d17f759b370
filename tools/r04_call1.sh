#!/bin/bash
# r04 call 1 (GPU box): what the headline is on a stream of distinct batches, before any kernel work of this round.
#   counters the box offers for the Infinity Cache, the rotating bench line + kernel stats + PMC passes (tools/profile_round.sh),
#   the filter-geometry gather with and without replay.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|hbm|EA0|EA_|TCC_.*(PROBE|STREAM|NC|UC|RW)" | head -200) > gpurun_out/r04/counters_list.txt 2>&1
tools/microbench/gather_bench filter 8 200000 12 0 4 > gpurun_out/r04/gather_filter.jsonl 2>&1
tools/microbench/gather_bench filter 8 200000 12 1 6 >> gpurun_out/r04/gather_filter.jsonl 2>&1
tools/microbench/gather_bench filter 8 200000 24 0 4 >> gpurun_out/r04/gather_filter.jsonl 2>&1
tools/microbench/gather_bench filter 8 200000 24 1 6 >> gpurun_out/r04/gather_filter.jsonl 2>&1
tools/microbench/gather_bench filter 8 200000 6 1 6 >> gpurun_out/r04/gather_filter.jsonl 2>&1
bash tools/profile_round.sh r04a
tail -c 1500 gpurun_out/prof/bench_r04a.json
