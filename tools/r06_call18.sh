#!/bin/bash
# r06: single-occurrence hits aligned from their own codes (Sel::key + record context): tests, then A/B of the headline
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_locate_topk.py tests/test_gpu_parity.py tests/test_gpu_capped.py tests/test_compact_host.py -m gpu -x -q > $O/pytest_c18.log 2>&1
tail -5 $O/pytest_c18.log
V=$GRAFT_REPO_ROOT/dicey_amd/variants
B="--steps 20 --warmup 8 --no-extra-configs --no-cpu-baseline --parity-queries 1000 --cli-queries 0"
timeout 900 python bench.py $B --keep-index --detail-out $O/dc_on.json > $O/dc_on.line 2> $O/dc_on.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
DICEY_LIB=$V/libdiceygpu_exp.so DICEY_NO_DIRECT_CTX=1 timeout 600 python bench.py --fm9 $FM9 $B --detail-out $O/dc_off.json > $O/dc_off.line 2> $O/dc_off.err
timeout 600 python bench.py --fm9 $FM9 $B --detail-out $O/dc_on2.json > $O/dc_on2.line 2> $O/dc_on2.err
DICEY_LIB=$V/libdiceygpu_exp.so DICEY_NO_DIRECT_CTX=1 timeout 600 python bench.py --fm9 $FM9 $B --detail-out $O/dc_off2.json > $O/dc_off2.line 2> $O/dc_off2.err
timeout 600 python bench.py --fm9 $FM9 $B --n-frac 0.05 --detail-out $O/dc_nmix.json > $O/dc_nmix.line 2> $O/dc_nmix.err
python - <<'PY'
import json
for n in ("on", "off", "on2", "off2", "nmix"):
    try:
        d = json.load(open("gpurun_out/r06/dc_%s.json" % n))
        s = d.get("sustained") or {}
        print("%-5s value %.1f M  ms/step %.4f  busy %.4f  sustained %.1f M  one-in-flight %.1f M  phases %s parity %s" % (
            n, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], s.get("value", 0) / 1e6, (d.get("value_one_in_flight") or {}).get("value", 0) / 1e6,
            {k: round(v, 4) for k, v in (d.get("phases_ms") or {}).items()}, d.get("parity_sample")))
    except Exception as e:
        print(n, "failed", e, open("gpurun_out/r06/dc_%s.err" % n).read()[-600:])
PY
rm -f /dev/shm/dicey_bench_*
