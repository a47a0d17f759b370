#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize_layout.py tests/test_gpu_parity.py tests/test_gpu_capped.py -m gpu -x -q > $O/pytest_n2.log 2>&1
tail -4 $O/pytest_n2.log
bash tools/kstats.sh r06nmix2 --n-frac 0.05 --no-extra-configs --parity-queries 0 --steps 10 --warmup 4 --in-flight 1 --keep-index 2>&1 | grep -E "k_search|k_nres|k_nkeep|k_walk|k_take|k_group_pack|k_verify|k_locate "
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
timeout 900 python bench.py --fm9 $FM9 --n-frac 0.05 --steps 30 --warmup 8 --no-extras --no-extra-configs --cpu-seconds 3 --parity-queries 300 \
  --detail-out $O/nmix_detail_i.json > $O/nmix_i.json 2> $O/nmix_i.err
timeout 900 python bench.py --fm9 $FM9 --n-frac 0.05 --hamming --steps 30 --warmup 8 --no-extras --no-extra-configs --cpu-seconds 3 --parity-queries 300 \
  --detail-out $O/nmixh_detail_i.json > $O/nmixh_i.json 2> $O/nmixh_i.err
python - <<'PY'
import json
for n in ("nmix", "nmixh"):
    d = json.load(open("gpurun_out/r06/%s_detail_i.json" % n))
    print(n, "%.1f M" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in d["phases_ms"].items()}, d.get("parity_sample"), d["roofline"]["kernel"])
PY
rm -f /dev/shm/dicey_bench_*
