#!/bin/bash
# r06: randomised differential runs against the checker in the full-size layout (table order 17 / 16, long filter 18: the layouts in
# which single-occurrence hits are aligned from their own codes) and on repeat-rich genomes -> profiles/r06_fuzz.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
BID=$(python -c "import bench; print(bench.build_id())")
(echo "# tools/r06_fuzz.sh: tools/fuzz_hunt.py in the full-size layouts (DICEY_KMER_K=17|16 DICEY_KMER_K2=18, hash-set checker neighbourhoods) and tools/fuzz_repeats.py, build $BID"
 for S in 21 22 23; do echo "## fuzz_hunt seed $S, K=17"; DICEY_KMER_K=17 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "configurations|MISMATCH|mismatch|Traceback|Error" | head -20; done
 for S in 24 25; do echo "## fuzz_hunt seed $S, K=16"; DICEY_KMER_K=16 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "configurations|MISMATCH|mismatch|Traceback|Error" | head -20; done
 for S in 31 32 33 34; do echo "## fuzz_repeats seed $S"; timeout 900 python tools/fuzz_repeats.py $S 10 2>&1 | grep -E "configurations|MISMATCH|mismatch|Traceback|Error|ok:" | head -20; done) > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
