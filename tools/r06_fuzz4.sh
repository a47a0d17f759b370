#!/bin/bash
# r06: the general fuzzers once more on the final sources (new seeds) -> profiles/r06_fuzz_final.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
BID=$(python -c "import bench; print(bench.build_id())")
F='configurations|MISMATCH|mismatch|got |want|Traceback|Error|refused'
(echo "# tools/r06_fuzz4.sh, build $BID"
 for S in 301 302 303 304; do echo "## fuzz_hunt seed $S, K=17 K2=18"; DICEY_KMER_K=17 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 311 312 313 314; do echo "## fuzz_hunt seed $S, K=16 K2=18"; DICEY_KMER_K=16 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 321 322 323; do echo "## fuzz_hunt seed $S, default layout"; FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 331 332 333 334; do echo "## fuzz_repeats seed $S"; timeout 900 python tools/fuzz_repeats.py $S 10 2>&1 | grep -E "$F|ok:" | head -20; done
 for S in 341 342 343 344; do echo "## fuzz_n seed $S, K=16 K2=19"; DICEY_KMER_K=16 DICEY_KMER_K2=19 timeout 600 python tools/fuzz_n.py $S 30 2>&1 | grep -E "$F" | head -20; done) > $O/fuzz_final.txt 2>&1
grep -E "##|failing|MISMATCH|got |want" $O/fuzz_final.txt | cut -c1-220
