#!/bin/bash
# A/B of k_search2p builds: each variant library in dicey_amd/variants/ takes the product library's place for one run of the list
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
cp dicey_amd/libdiceygpu.so /tmp/libdiceygpu_product.so
for V in $VARIANTS; do
  cp dicey_amd/variants/libdiceygpu_$V.so dicey_amd/libdiceygpu.so
  echo "== variant $V"
  bash tools/r05_exp.sh 8_$V tools/r05_exp8.list 2>&1 | grep -E "^d[12]|^d2no|^search|^padlock|k_search2p<true, true> (TCP_TOTAL|TCP_TCC_READ|TCC_EA0_RDREQ_sum|SQ_INSTS_VALU|SQ_INSTS_SALU|TCC_HIT)"
done
cp /tmp/libdiceygpu_product.so dicey_amd/libdiceygpu.so
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest $TESTS -x -q > gpurun_out/r05/pytest_k.log 2>&1; tail -5 gpurun_out/r05/pytest_k.log; fi
