#!/bin/bash
# r06: the whole GPU suite, then the N-mix configuration and the repeats configuration
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
T=${1:-e}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_full_$T.log 2>&1
tail -18 $O/pytest_full_$T.log
timeout 900 python bench.py --n-frac 0.05 --steps 30 --warmup 8 --no-extras --no-extra-configs --cpu-seconds 3 --parity-queries 300 --keep-index \
  --detail-out $O/nmix_detail_$T.json > $O/nmix_$T.json 2> $O/nmix_$T.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 600 python bench.py --config hunt_d2 --n-frac 0.05 --fm9 $FM9 --steps 10 --warmup 4 --no-extras --no-extra-configs --cpu-seconds 2 --parity-queries 200 \
  --detail-out $O/nmix2_detail_$T.json > $O/nmix2_$T.json 2> $O/nmix2_$T.err
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --genome repeats --steps 30 --warmup 8 --no-extras --no-extra-configs --cpu-seconds 3 --parity-queries 300 \
  --detail-out $O/repeats_detail_$T.json > $O/repeats_$T.json 2> $O/repeats_$T.err
python - $T <<'PY'
import json, sys
for n in ("nmix", "nmix2", "repeats"):
    try:
        d = json.load(open("gpurun_out/r06/%s_detail_%s.json" % (n, sys.argv[1])))
        print(n, "%.1f M" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in d["phases_ms"].items()}, d.get("parity_sample"), d["roofline"]["kernel"])
    except Exception as e:
        print(n, "failed", e)
PY
rm -f /dev/shm/dicey_bench_*
