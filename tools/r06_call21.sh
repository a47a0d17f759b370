#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_padlock.py tests/test_gpu_thal_wave.py -m gpu -x -q 2>&1 | tail -2
for S in 421 422 423; do timeout 300 python tools/fuzz_padlock.py $S 2>&1 | grep -E "failing|MISMATCH|Traceback|Error" | head -3; done
DICEY_TIMING=1 timeout 800 python bench.py --config padlock --steps 4 --warmup 2 --no-cpu-baseline --parity-queries 14 --no-extra-configs > $O/pad_t2.line 2> $O/pad_t2.err
grep -E "dg_padlock_scan" $O/pad_t2.err | tail -12
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06/pad_t2.line") if l.startswith("{")][-1])
print("padlock %.0f genes/s  %.1f ms/step  parity %s" % (d["value"], d["ms_per_step"], d.get("parity_sample")))
PY
rm -f /dev/shm/dicey_bench_*
