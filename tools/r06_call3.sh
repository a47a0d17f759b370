#!/bin/bash
# r06: phase clocks of the locate job kernels (development build with -DDG_TOPK_PROFILE) on the repeats genome, one batch at a time
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python bench.py --genome repeats --steps 4 --warmup 3 --no-extras --no-extra-configs --no-cpu-baseline --parity-queries 0 --keep-index --in-flight 1 \
  --detail-out $O/prof_build_detail.json > $O/prof_build.json 2> $O/prof_build.err
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
DICEY_LIB=$GRAFT_REPO_ROOT/dicey_amd/variants/libdiceygpu_prof.so timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 4 --warmup 3 --no-extras --no-extra-configs \
  --no-cpu-baseline --parity-queries 0 --in-flight 1 --detail-out $O/prof_detail.json > $O/prof.json 2> $O/prof.err
grep "topk profile" $O/prof.err | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/prof_detail.json"))
print("prof build:", d["value"], d["phases_ms"])
PY
rm -f /dev/shm/dicey_bench_*
