#!/bin/bash
# r03 call 12: cleaned build (index-derived context removed, select by one or four wavefronts, sequence starts in LDS): whole suite,
# the driver's default command, repeats and d2 lines
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03n
rm -rf $OUT; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
( time timeout 1200 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r03n/bench.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print("default:", round(d["value"]), d["ms_per_step"], {k: round(v, 4) for k, v in d["phases_ms"].items()}, d.get("parity_sample"))
        print(" d2h:", d.get("value_with_d2h"), "\n h2h:", d.get("host_to_host_pipelined"), "\n cli:", (d.get("cli_end_to_end") or {}).get("value"), (d.get("cli_end_to_end_after_release") or {}).get("value"))
        print(" cpu:", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline_parallel") or {}).get("value"))
        for k, v in ((kk, vv) for kk, vv in (d.get("extra_configs") or {}).items() if isinstance(vv, dict)):
            print(k, {a: v.get(a) for a in ("value", "unit", "ms_per_step", "phases_ms", "parity_sample", "error", "wall_s")})
PY
