#!/bin/bash
# lanes experiment: libraries with 1 + DG_NEXTRA lanes (dicey_amd/variants/libdiceygpu_n<N>.so) against the product library, bench --in-flight
# usage: GENOME=iid|repeats bash tools/r05_call13.sh
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
G=${GENOME:-iid}
cp dicey_amd/libdiceygpu.so /tmp/libdiceygpu_product.so
rm -f /dev/shm/dicey_bench_*
timeout 900 python bench.py --genome $G --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-extra-configs --parity-queries 0 --keep-index --detail-out /tmp/b.json > /dev/null 2> gpurun_out/r05/lanes_build.err
FM9=$(ls /dev/shm/dicey_bench_*.fm9 | head -1)
echo "index $FM9"
run() {  # name, in-flight
  timeout 600 python bench.py --genome $G --fm9 $FM9 --no-extras --no-extra-configs --no-cpu-baseline --parity-queries 0 --steps 40 --warmup 8 --in-flight $2 --detail-out gpurun_out/r05/lanes_$1.json > /dev/null 2> gpurun_out/r05/lanes_$1.err
  python -c "import json; j=json.load(open('gpurun_out/r05/lanes_$1.json')); print('$1', round(j['value']/1e6,2), 'M', round(j['ms_per_step'],4), 'ms')"
}
for R in 1 2; do
  run f3 3
  cp dicey_amd/variants/libdiceygpu_n5.so dicey_amd/libdiceygpu.so
  run f4 4; run f6 6
  cp /tmp/libdiceygpu_product.so dicey_amd/libdiceygpu.so
done
