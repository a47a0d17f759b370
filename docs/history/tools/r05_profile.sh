#!/bin/bash
# GPU box: the profile round of r05 — default bench line + kernel trace + PMC passes (tools/profile_round.sh), the thal counters of the
# search / padlock configurations (tools/prof_thal.sh), kernel stats + memory counters of distance 2 and of the repeats genome
# (tools/prof_cfg.sh), the kernel-shaped gather microbenchmark.  Summaries: tools/summarize_profile.py (run where the repo lives).
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
tail -3 gpurun_out/profile_round_r05.log
for L in 12 15; do tools/microbench/gather_bench filter2 27 200000 $L 6; done > gpurun_out/prof/gather_filter2.jsonl 2>&1
bash tools/prof_thal.sh > gpurun_out/prof_thal_r05.log 2>&1
tail -2 gpurun_out/prof_thal_r05.log | cut -c1-300
bash tools/prof_cfg.sh r05d2 --config hunt_d2 > gpurun_out/prof_cfg_r05d2.log 2>&1
tail -6 gpurun_out/prof_cfg_r05d2.log
rm -f /dev/shm/dicey_bench_*
bash tools/prof_cfg.sh r05rep --genome repeats > gpurun_out/prof_cfg_r05rep.log 2>&1
tail -12 gpurun_out/prof_cfg_r05rep.log
rm -f /dev/shm/dicey_bench_*
