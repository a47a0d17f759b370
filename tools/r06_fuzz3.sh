#!/bin/bash
# r06: tools/fuzz_n.py — texts full of N runs and sequence boundaries — in three layouts -> profiles/r06_fuzz_n.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
BID=$(python -c "import bench; print(bench.build_id())")
F='configurations|MISMATCH|got |want|Traceback|Error|refused'
(echo "# tools/r06_fuzz3.sh, build $BID"
 for S in 101 102 103 104 105 106 107 108; do echo "## fuzz_n seed $S, K=16 K2=18"; DICEY_KMER_K=16 DICEY_KMER_K2=18 timeout 600 python tools/fuzz_n.py $S 24 2>&1 | grep -E "$F" | head -20; done
 for S in 111 112 113 114 115 116; do echo "## fuzz_n seed $S, K=17 K2=18"; DICEY_KMER_K=17 DICEY_KMER_K2=18 timeout 600 python tools/fuzz_n.py $S 24 2>&1 | grep -E "$F" | head -20; done
 for S in 121 122 123 124 125 126; do echo "## fuzz_n seed $S, default layout"; timeout 600 python tools/fuzz_n.py $S 24 2>&1 | grep -E "$F" | head -20; done) > $O/fuzz_n.txt 2>&1
cat $O/fuzz_n.txt | cut -c1-260
