#!/bin/bash
# An A/B build of libgpmi.so with extra compiler flags: tools/build_variant.sh <tag> <flags...>  ->  tools/bin/libgpmi_<tag>.so (git-ignored; it
# travels to the GPU box).  Never the product library: a GPU script copies it over gaussianprocesses.jl_amd/lib/libgpmi.so on the box only.
set -e
TAG=$1; shift
cd "$(dirname "$0")/../gaussianprocesses.jl_amd/csrc"
O=/tmp/gpmi_variant_$TAG; mkdir -p $O ../../tools/bin
FL="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value $*"
for f in api cov gemm update256 panel chain grad fitc dev_hip; do
  X=""; case $f in gemm|update256) X="-mllvm -amdgpu-atomic-optimizer-strategy=None";; esac
  /opt/rocm/bin/hipcc $FL $X -c $f.hip -o $O/$f.o &
done
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall "$@" -x c++ -c blocked.cpp -o $O/blocked.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.o -ldl -pthread -Wl,--version-script=libgpmi.map -o ../../tools/bin/libgpmi_$TAG.so
echo built tools/bin/libgpmi_$TAG.so
