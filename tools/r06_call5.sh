#!/bin/bash
# r06: (1) phase clocks of the locate job kernels again; (2) verify with its phases switched off (DICEY_EXP) and CH 4 / 8;
# (3) the default configuration: as is, with staggered first submissions, with lane stream priorities
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
B="--no-extras --no-extra-configs --no-cpu-baseline --parity-queries 0"
timeout 900 python bench.py --genome repeats --steps 4 --warmup 3 $B --keep-index --in-flight 1 --detail-out $O/x.json > /dev/null 2> $O/e5_build.err
FM9=$(ls /dev/shm/dicey_bench_*repeats*.fm9 | head -1)
DICEY_LIB=$GRAFT_REPO_ROOT/dicey_amd/variants/libdiceygpu_prof.so timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 4 --warmup 3 $B --in-flight 1 --detail-out $O/x.json > /dev/null 2> $O/prof5.err
grep "topk profile" $O/prof5.err | tail -1
for V in "0:8" "16:8" "32:8" "48:8" "0:4" "16:4" "48:4"; do
  E=${V%%:*}; CH=${V##*:}
  DICEY_EXP=$E DICEY_VERIFY_CH=$CH timeout 600 python bench.py --genome repeats --fm9 $FM9 --steps 10 --warmup 4 $B --in-flight 1 --detail-out $O/v_${E}_$CH.json > /dev/null 2> $O/v.err
  python - $E $CH <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06/v_%s_%s.json" % (sys.argv[1], sys.argv[2])))
print("verify EXP=%s CH=%s: ms_verify %.3f  ms_total %.3f" % (sys.argv[1], sys.argv[2], d["phases_ms"]["ms_verify"], d["phases_ms"]["ms_total"]))
PY
done
rm -f /dev/shm/dicey_bench_*repeats*
# ---- default configuration: lanes
timeout 900 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --parity-queries 0 --cli-queries 0 --keep-index --detail-out $O/lanes_base.json > $O/lanes_base.line 2> $O/lanes_base.err
FM9=$(ls /dev/shm/dicey_bench_*iid*.fm9 | head -1)
for V in "base2:" "st60:--stagger-us 60" "st120:--stagger-us 120" "prio:"; do
  N=${V%%:*}; A=${V#*:}
  P=""; [ $N = prio ] && P="DICEY_EXP_PRIO=1"
  env $P timeout 600 python bench.py --fm9 $FM9 --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --parity-queries 0 --cli-queries 0 $A --detail-out $O/lanes_$N.json > $O/lanes_$N.line 2> $O/lanes_$N.err
done
python - <<'PY'
import json
for n in ("base", "base2", "st60", "st120", "prio"):
    try:
        d = json.load(open("gpurun_out/r06/lanes_%s.json" % n))
        s = d.get("sustained") or {}
        print("%-6s value %.1f M  ms/step %.4f  kernel_ms(busy) %.4f launch_ms %.4f  sustained %.1f M over %.2f s  one-at-a-time %.1f M" % (
            n, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["launch_ms"], s.get("value", 0) / 1e6, s.get("seconds", 0),
            d.get("value_one_in_flight", {}).get("value", 0) / 1e6))
    except Exception as e:
        print(n, "failed", e)
PY
rm -f /dev/shm/dicey_bench_*
