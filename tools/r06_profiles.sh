#!/bin/bash
# r06, the round's profile call: default configuration (bench line, kernel trace + stats, PMC passes), edit distance 2, 5 % N, Hamming 2, the thal
# kernels, then the repeats genome.  tools/summarize_profile.py turns gpurun_out/prof* into profiles/ afterwards.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
KEEP_INDEX=1 bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
tail -2 gpurun_out/profile_round_r06.log
bash tools/prof_cfg.sh d2 --config hunt_d2 > gpurun_out/prof_cfg_r06d2.log 2>&1
tail -3 gpurun_out/prof_cfg_r06d2.log
bash tools/prof_cfg.sh nmix --config hunt_d1 --n-frac 0.05 > gpurun_out/prof_cfg_r06nmix.log 2>&1
tail -3 gpurun_out/prof_cfg_r06nmix.log
bash tools/prof_cfg.sh ham2 --config hunt_d2 --hamming > gpurun_out/prof_cfg_r06ham2.log 2>&1
tail -3 gpurun_out/prof_cfg_r06ham2.log
bash tools/prof_thal.sh > gpurun_out/prof_thal_r06.log 2>&1
tail -c 600 gpurun_out/prof_thal_r06.log
rm -f /dev/shm/dicey_bench_*
bash tools/prof_cfg.sh rep --genome repeats > gpurun_out/prof_cfg_r06rep.log 2>&1
tail -12 gpurun_out/prof_cfg_r06rep.log
rm -f /dev/shm/dicey_bench_*
find gpurun_out/prof gpurun_out/prof_d2 gpurun_out/prof_nmix gpurun_out/prof_ham2 gpurun_out/prof_rep gpurun_out/prof_thal -name "*.db" -delete 2>/dev/null
du -sh gpurun_out/prof gpurun_out/prof_d2 gpurun_out/prof_nmix gpurun_out/prof_ham2 gpurun_out/prof_rep gpurun_out/prof_thal
