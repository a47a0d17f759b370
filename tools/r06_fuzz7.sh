#!/bin/bash
# r06: the last sources once more: fuzz_padlock (the scan's host side changed last), a few seeds of the others -> profiles/r06_fuzz_last.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
BID=$(python -c "import bench; print(bench.build_id())")
F='configurations|MISMATCH|mismatch|got |want|Traceback|Error|refused'
(echo "# tools/r06_fuzz7.sh, build $BID"
 for S in 601 602 603 604 605 606; do echo "## fuzz_padlock seed $S"; timeout 300 python tools/fuzz_padlock.py $S 2>&1 | grep -E "$F" | head -6; done
 for S in 611 612; do echo "## fuzz_hunt seed $S, K=16 K2=18"; DICEY_KMER_K=16 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 400 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -10; done
 for S in 621 622; do echo "## fuzz_n seed $S, K=17 K2=18"; DICEY_KMER_K=17 DICEY_KMER_K2=18 timeout 400 python tools/fuzz_n.py $S 30 2>&1 | grep -E "$F" | head -10; done
 for S in 631; do echo "## fuzz_search seed $S"; timeout 300 python tools/fuzz_search.py $S 2>&1 | grep -E "$F" | head -6; done) > $O/fuzz_last.txt 2>&1
cat $O/fuzz_last.txt | cut -c1-200
