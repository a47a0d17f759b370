#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
bash tools/kstats.sh r06nmix --n-frac 0.05 --no-extra-configs --parity-queries 0 --steps 10 --warmup 4 --in-flight 1 2>&1 | grep -v "k_kf\|k_pre5\|k_plv\|k_derive\|k_kmer\|k_decode\|k_block\|k_ktab\|k_check\|k_nrun\|k_chunk\|k_selfcheck"
rm -f /dev/shm/dicey_bench_*
