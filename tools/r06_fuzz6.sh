#!/bin/bash
# r06: fuzz_hunt / fuzz_n on an index opened the way `dicey hunt` opens it (DG_OPEN_COMPACT | DG_OPEN_NO_PRE5) -> profiles/r06_fuzz_one_shot.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp FUZZ_ONE_SHOT=1
O=gpurun_out/r06
mkdir -p $O
BID=$(python -c "import bench; print(bench.build_id())")
F='configurations|MISMATCH|mismatch|got |want|Traceback|Error|refused'
(echo "# tools/r06_fuzz6.sh (FUZZ_ONE_SHOT=1), build $BID"
 for S in 501 502; do echo "## fuzz_hunt seed $S, K=17 K2=18"; DICEY_KMER_K=17 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 511 512; do echo "## fuzz_hunt seed $S, K=16 K2=18"; DICEY_KMER_K=16 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 521 522; do echo "## fuzz_hunt seed $S, default layout"; FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 531 532 533; do echo "## fuzz_n seed $S, K=16 K2=18"; DICEY_KMER_K=16 DICEY_KMER_K2=18 timeout 600 python tools/fuzz_n.py $S 30 2>&1 | grep -E "$F" | head -20; done
 for S in 541 542; do echo "## fuzz_repeats-like: fuzz_n seed $S, default layout"; timeout 600 python tools/fuzz_n.py $S 30 2>&1 | grep -E "$F" | head -20; done) > $O/fuzz_one_shot.txt 2>&1
cat $O/fuzz_one_shot.txt | cut -c1-220
