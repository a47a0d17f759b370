#!/bin/bash
# r06: N-mix sub-line, N tests, then the default run exactly as the driver starts it
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
T=${1:-g}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize_layout.py::test_queries_with_n_where_the_text_has_no_short_n_run "tests/test_gpu_parity.py::test_every_search_mode_gives_the_same_hits" -m gpu -x -q > $O/pytest_n_$T.log 2>&1
tail -3 $O/pytest_n_$T.log
timeout 900 python bench.py --n-frac 0.05 --steps 30 --warmup 8 --no-extras --no-extra-configs --cpu-seconds 3 --parity-queries 300 \
  --detail-out $O/nmix_detail_$T.json > $O/nmix_$T.json 2> $O/nmix_$T.err
python - $T <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06/nmix_detail_%s.json" % sys.argv[1]))
print("nmix", "%.1f M" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in d["phases_ms"].items()}, d.get("parity_sample"), d["roofline"]["kernel"])
PY
rm -f /dev/shm/dicey_bench_*
( time timeout 1500 python bench.py --steps 20 --warmup 5 --detail-out $O/default_detail_$T.json ) > $O/default_$T.json 2> $O/default_$T.err
tail -1 $O/default_$T.json | cut -c1-3900
tail -4 $O/default_$T.err
rm -f /dev/shm/dicey_bench_*
