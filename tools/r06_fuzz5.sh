#!/bin/bash
# r06: fuzz_search / fuzz_padlock with more seeds on the final sources -> profiles/r06_fuzz_search_padlock.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
BID=$(python -c "import bench; print(bench.build_id())")
(echo "# tools/r06_fuzz5.sh, build $BID"
 for S in 401 402 403 404 405 406; do echo "## fuzz_search seed $S"; timeout 300 python tools/fuzz_search.py $S 2>&1 | grep -E "failing|MISMATCH|mismatch|Traceback|Error" | head -8; done
 for S in 411 412 413 414 415 416; do echo "## fuzz_padlock seed $S"; timeout 300 python tools/fuzz_padlock.py $S 2>&1 | grep -E "failing|MISMATCH|mismatch|Traceback|Error" | head -8; done) > $O/fuzz_sp.txt 2>&1
cat $O/fuzz_sp.txt | cut -c1-200
