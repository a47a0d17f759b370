#!/bin/bash
# r06: a wider randomised campaign (more seeds, three layouts, d = 2 with literal checker neighbourhoods on a few) -> profiles/r06_fuzz_wide.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
BID=$(python -c "import bench; print(bench.build_id())")
F='configurations|MISMATCH|mismatch|Traceback|Error|refused'
(echo "# tools/r06_fuzz2.sh, build $BID"
 for S in 41 42 43 44 45 46 47 48; do echo "## fuzz_hunt seed $S, K=17 K2=18"; DICEY_KMER_K=17 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 51 52 53 54 55 56; do echo "## fuzz_hunt seed $S, K=16 K2=18"; DICEY_KMER_K=16 DICEY_KMER_K2=18 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 61 62 63 64 65 66; do echo "## fuzz_hunt seed $S, default layout"; FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 71 72; do echo "## fuzz_hunt seed $S, K=16 K2=19"; DICEY_KMER_K=16 DICEY_KMER_K2=19 FUZZ_FAST_NEIGHBORS=1 timeout 600 python tools/fuzz_hunt.py $S 40 2>&1 | grep -E "$F" | head -20; done
 for S in 81 82 83 84 85 86; do echo "## fuzz_repeats seed $S"; timeout 900 python tools/fuzz_repeats.py $S 10 2>&1 | grep -E "$F|ok:" | head -20; done
 for S in 91 92; do echo "## fuzz_search seed $S"; timeout 600 python tools/fuzz_search.py $S 2>&1 | tail -3; done
 for S in 95 96; do echo "## fuzz_padlock seed $S"; timeout 600 python tools/fuzz_padlock.py $S 2>&1 | tail -3; done) > $O/fuzz_wide.txt 2>&1
cat $O/fuzz_wide.txt | cut -c1-220
