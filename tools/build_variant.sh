#!/bin/bash
# Development build of libdiceygpu.so with extra compiler flags -> dicey_amd/variants/libdiceygpu_<name>.so (git-ignored; travels with
# gpurun; DICEY_LIB=<path> makes the Python mirror load it).  usage: tools/build_variant.sh <name> <flags...>
set -e
cd "$(dirname "$0")/../dicey_amd/csrc"
N=$1; shift
mkdir -p ../variants build_$N
for f in index seam hunt search build thal_api padlock; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off "$@" -c $f.hip -o build_$N/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libdiceygpu_$N.so build_$N/*.o
rm -rf build_$N
echo built ../variants/libdiceygpu_$N.so
