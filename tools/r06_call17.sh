#!/bin/bash
# r06: DG_OPEN_NO_PRE5 — tests, what the distance-1 / distance-2 step costs without the array, the binary's 10 M run
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q > $O/pytest_c17.log 2>&1
tail -3 $O/pytest_c17.log
V=$GRAFT_REPO_ROOT/dicey_amd/variants
B="--steps 20 --warmup 8 --no-extra-configs --no-cpu-baseline --parity-queries 300 --cli-queries 0 --no-extras"
timeout 900 python bench.py $B --keep-index --detail-out $O/np_base.json > $O/np_base.line 2> $O/np_base.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
DICEY_LIB=$V/libdiceygpu_exp.so DICEY_NO_PRE5=1 timeout 600 python bench.py --fm9 $FM9 $B --detail-out $O/np_nopre5.json > $O/np_nopre5.line 2> $O/np_nopre5.err
timeout 600 python bench.py --fm9 $FM9 $B --config hunt_d2 --detail-out $O/np_d2.json > $O/np_d2.line 2> $O/np_d2.err
DICEY_LIB=$V/libdiceygpu_exp.so DICEY_NO_PRE5=1 timeout 600 python bench.py --fm9 $FM9 $B --config hunt_d2 --detail-out $O/np_d2_nopre5.json > $O/np_d2_nopre5.line 2> $O/np_d2_nopre5.err
python - <<'PY'
import json
for n in ("base", "nopre5", "d2", "d2_nopre5"):
    try:
        d = json.load(open("gpurun_out/r06/np_%s.json" % n))
        print("%-10s value %.1f M  ms/step %.4f  busy %.4f  parity %s" % (n, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("parity_sample")))
    except Exception as e:
        print(n, "failed", e, open("gpurun_out/r06/np_%s.err" % n).read()[-600:])
PY
timeout 900 python tools/cli_10m.py $FM9 10000000 3 2>&1 | grep -v amdgpu.ids | tee $O/cli_10m_c.txt
rm -f /dev/shm/dicey_bench_*
