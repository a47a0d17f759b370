#!/bin/bash
# r03 call 13: packed fetch + pinned upload: parity / CLI / multirank modules, default line with its delivery extras
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03o
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_multirank.py tests/test_gpu_capped.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 900 python bench.py --no-extra-configs --cpu-seconds 3 > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r03o/bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("default:", round(d["value"]), d["ms_per_step"], {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"))
        print(" d2h:", d.get("value_with_d2h"), "\n h2h:", d.get("host_to_host_pipelined"))
PY
