#!/bin/bash
# GPU box: wave thal kernel vs sequential kernel on the 3.1 Gb search workload (every hit's Tm and position), timing, tests.
cd "$GRAFT_REPO_ROOT"
timeout 200 python bench.py --keep-index --steps 2 --warmup 1 --no-cpu-baseline > /tmp/b1.log 2>&1
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
if [ "${1:-}" != "fast" ]; then
DICEY_DEBUG_DUMP_RAW=/tmp/s.raw DICEY_DEBUG_THAL_REDO=1 timeout 300 python tools/bench_search.py --no-cpu --fm9 $FM9 > /dev/null 2>&1
fi
DICEY_DEBUG_DUMP_RAW=/tmp/w.raw timeout 300 python tools/bench_search.py --no-cpu --fm9 $FM9 2>&1 | grep -E "^\{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('seconds','thal_per_s','sites','ms_device')})"
if [ "${1:-}" != "fast" ]; then
python tools/diff_raw.py /tmp/w.raw /tmp/s.raw | grep -E "alignpos|temp|bad"
timeout 600 python -m pytest tests -m gpu -x -q -k "thal or search" 2>&1 | tail -3
fi
rm -f /dev/shm/dicey_bench_*
