#!/bin/bash
# r04 call 15: where the select-inside form of k_search2p loses its time: search only (select skipped), counters of both forms.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index > /dev/null 2>&1
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
for v in fused skip generic; do
  unset DICEY_NO_FUSED_SELECT2 DICEY_EXP_SKIP_SELECT2
  if [ $v = generic ]; then export DICEY_NO_FUSED_SELECT2=1; fi
  if [ $v = skip ]; then export DICEY_EXP_SKIP_SELECT2=1; fi
  timeout 600 python bench.py --fm9 $FM9 --config hunt_d2 --steps 6 --warmup 2 --no-cpu-baseline --parity-queries 0 --no-extras --no-extra-configs --in-flight 1 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('d2 $v', round(j['value']/1e6,2), j['ms_per_step'], j['phases_ms'], j['hits_per_step'], j['leaves_per_step'])"
done
for v in fused generic; do
  unset DICEY_NO_FUSED_SELECT2 DICEY_EXP_SKIP_SELECT2
  if [ $v = generic ]; then export DICEY_NO_FUSED_SELECT2=1; fi
  i=0
  while read -r C; do
    [ -z "$C" ] && continue
    i=$((i+1))
    (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r04/pmc15_${v}_$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --fm9 $FM9 --config hunt_d2 --no-cpu-baseline --no-extras --no-extra-configs --parity-queries 0 --steps 2 --warmup 1 --in-flight 1 > /dev/null 2>&1)
  done <<LIST
SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
LIST
  python - $v <<'PY'
import csv, glob, sys
v=sys.argv[1]; acc={}
for d in sorted(glob.glob(f"gpurun_out/r04/pmc15_{v}_*/**/pmc_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(d)):
        if "k_search2p" not in r["Kernel_Name"]: continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(v, {k: round(sum(x)/len(x)/1e6,1) for k,x in sorted(acc.items())})
PY
done
rm -rf gpurun_out/r04/pmc15_*/*/*.db
rm -f /dev/shm/dicey_bench_*
