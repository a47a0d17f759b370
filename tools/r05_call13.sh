#!/bin/bash
# lanes experiment: libraries with 1 + DG_NEXTRA lanes (dicey_amd/variants/libdiceygpu_n<N>.so) against the product library, bench --in-flight
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
cp dicey_amd/libdiceygpu.so /tmp/libdiceygpu_product.so
cat > /tmp/l3.list <<'L'
d1f3|X=1|--steps 90 --warmup 12 --parity-queries 0 --in-flight 3|0
d2f3|X=1|--config hunt_d2 --steps 40 --warmup 8 --parity-queries 0 --in-flight 3|0
L
cat > /tmp/l4.list <<'L'
d1f4|X=1|--steps 90 --warmup 12 --parity-queries 0 --in-flight 4|0
d2f4|X=1|--config hunt_d2 --steps 40 --warmup 8 --parity-queries 0 --in-flight 4|0
L
cat > /tmp/l6.list <<'L'
d1f5|X=1|--steps 90 --warmup 12 --parity-queries 0 --in-flight 5|0
d1f6|X=1|--steps 90 --warmup 12 --parity-queries 0 --in-flight 6|0
d2f6|X=1|--config hunt_d2 --steps 40 --warmup 8 --parity-queries 0 --in-flight 6|0
L
for R in 1 2; do
bash tools/r05_exp.sh 13 /tmp/l3.list 2>&1 | grep -E "^d[12]"
cp dicey_amd/variants/libdiceygpu_n3.so dicey_amd/libdiceygpu.so
bash tools/r05_exp.sh 13 /tmp/l4.list 2>&1 | grep -E "^d[12]"
cp dicey_amd/variants/libdiceygpu_n5.so dicey_amd/libdiceygpu.so
bash tools/r05_exp.sh 13 /tmp/l6.list 2>&1 | grep -E "^d[12]"
cp /tmp/libdiceygpu_product.so dicey_amd/libdiceygpu.so
done
