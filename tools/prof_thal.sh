#!/bin/bash
# GPU box: fp64 / LDS instruction counters of the thal() wave kernels on the `search` and `padlock` bench configurations
# (place k_site_wave / k_thal_self_wave against the fp64 vector peak and the LDS).  Separate --pmc passes,
# kernel trace only.  Summary -> gpurun_out/prof_thal/thal_counters.json, stamped with the sources' build_id; copy it to profiles/thal_counters.json.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof_thal
rm -rf $OUT; mkdir -p $OUT
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 2>/dev/null | head -1)
if [ -z "$FM9" ]; then
  timeout 600 python bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index --detail-out /tmp/build_detail.json > $OUT/build.json 2> $OUT/build.err
  FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
fi
for CFG in search padlock; do
  timeout 600 python bench.py --config $CFG --fm9 $FM9 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --detail-out $OUT/bench_$CFG.json > $OUT/bench_line_$CFG.json 2> $OUT/bench_$CFG.err
  i=0
  while read -r C; do
    [ -z "$C" ] && continue
    i=$((i+1))
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/${CFG}_pmc_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --fm9 $FM9 --no-cpu-baseline --no-extras --steps 1 --warmup 1 --detail-out /tmp/thal_pmc_detail.json > $GRAFT_REPO_ROOT/$OUT/${CFG}_pmc_$i.json 2> $GRAFT_REPO_ROOT/$OUT/${CFG}_pmc_$i.err)
  done <<LIST
SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
LIST
done
python - "$OUT" <<'PY'
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'tools'))
from profnames import short_kernel_name
out = sys.argv[1]
res = {}
for cfg in ("search", "padlock"):
    acc = {}
    for d in sorted(glob.glob(os.path.join(out, cfg + "_pmc_*", "pmc_counter_collection.csv"))):
        for r in csv.DictReader(open(d)):
            k = short_kernel_name(r["Kernel_Name"])
            if "wave" not in k or "kmer_table" in k: continue
            acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    ks = {}
    for (k, c), v in acc.items():
        ks.setdefault(k, {})[c] = max(v)  # the timed launch is the largest one (warm-up batches are the same size; max is stable)
    bench = None
    try:
        bench = json.load(open(os.path.join(out, "bench_%s.json" % cfg)))
    except Exception:
        pass
    res[cfg] = {"kernels": ks, "bench": {k: bench.get(k) for k in ("value", "unit", "ms_per_step", "phases_ms", "site_stage", "arm_thal_per_step", "probe_thal_per_step")} if bench else None}
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench as B
res["build_id"] = B.build_id()
json.dump(res, open(os.path.join(out, "thal_counters.json"), "w"), indent=1)
print(json.dumps(res)[:3000])
PY
rm -rf $OUT/*_pmc_*/*/*.db 2>/dev/null
