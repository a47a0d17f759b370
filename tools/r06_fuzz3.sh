#!/bin/bash
# r06: tools/fuzz_n.py — texts full of N runs and sequence boundaries — in three layouts -> profiles/r06_fuzz_n2.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
BID=$(python -c "import bench; print(bench.build_id())")
F='configurations|MISMATCH|got |want|Traceback|Error|refused'
(echo "# tools/r06_fuzz3.sh, build $BID"
 for S in 201 202 203 204 205 206 207 208; do echo "## fuzz_n seed $S, K=16 K2=18"; DICEY_KMER_K=16 DICEY_KMER_K2=18 timeout 600 python tools/fuzz_n.py $S 30 2>&1 | grep -E "$F" | head -20; done
 for S in 211 212 213 214 215 216; do echo "## fuzz_n seed $S, K=17 K2=18"; DICEY_KMER_K=17 DICEY_KMER_K2=18 timeout 600 python tools/fuzz_n.py $S 30 2>&1 | grep -E "$F" | head -20; done
 for S in 221 222 223 224 225 226; do echo "## fuzz_n seed $S, default layout"; timeout 600 python tools/fuzz_n.py $S 30 2>&1 | grep -E "$F" | head -20; done) > $O/fuzz_n2.txt 2>&1
cat $O/fuzz_n2.txt | cut -c1-260
