#!/bin/bash
# Builds A/B variants of libdiceygpu.so that differ in hunt.o only: tools/r05_variants.sh "name:rb:waves:extra compiler flags" ...
# (rb, waves: k_search2p's rounds per wait and wavefronts per SIMD; flags e.g. -DDICEY_S1_EXECLOAD=1)
# -> dicey_amd/variants/libdiceygpu_<name>.so (git-ignored; travels with gpurun); tools/r05_call10.sh runs them in turn.
set -e
cd "$(dirname "$0")/../dicey_amd/csrc"
W=/tmp/dicey_variants/a/b; rm -rf /tmp/dicey_variants; mkdir -p $W /tmp/dicey_variants/include ../variants
cp ../../include/*.h /tmp/dicey_variants/include/
for v in "$@"; do
  IFS=: read -r n rb w fl <<< "$v"
  cp *.hpp *.hip $W/
  sed -i "s/static constexpr u32 LONG2_RB = 2;/static constexpr u32 LONG2_RB = $rb;/; s/__launch_bounds__(256, 6) k_search2p/__launch_bounds__(256, $w) k_search2p/" $W/hunt_search.hpp
  (cd $W && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off $fl -Rpass-analysis=kernel-resource-usage -c hunt.hip -o hunt_$n.o 2>&1 | grep -A9 "Function Name: .*k_search1sILb1ELb1" | grep -E "VGPRs:|Scratch" )
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libdiceygpu_$n.so build/index.o build/seam.o $W/hunt_$n.o build/search.o build/build.o build/thal_api.o build/padlock.o
  echo built $n
done
