#!/bin/bash
# r04 call 21: profile round of the final build (three batches in flight).
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/profile_round.sh r04d > gpurun_out/profile_round_d.log 2>&1
tail -2 gpurun_out/profile_round_d.log
