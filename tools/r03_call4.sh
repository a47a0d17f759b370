#!/bin/bash
# r03 call 4: whole -m gpu suite + the default bench line (with extra_configs) of the tree as it stands
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03f
rm -rf $OUT; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r03f/bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("default:", d["value"], d["ms_per_step"], d["phases_ms"], d.get("parity_sample"))
        for k, v in ((kk, vv) for kk, vv in (d.get("extra_configs") or {}).items() if isinstance(vv, dict)):
            print(k, {a: v.get(a) for a in ("value", "unit", "ms_per_step", "phases_ms", "parity_sample", "error", "wall_s")})
PY
