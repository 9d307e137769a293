#!/bin/bash
# The library of tools/chain_trace.py: libgpmi.so's own objects with chain.hip recompiled under -DGPMI_CHAIN_TRACE (clock marks per task of
# the persistent chain kernel).  Output: tools/bin/libgpmi_chain_trace.so (git-ignored; it travels to the GPU box).  Never the product library.
set -e
cd "$(dirname "$0")/../gaussianprocesses.jl_amd/csrc"
make -s -j8
mkdir -p ../../tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -DGPMI_CHAIN_TRACE -c chain.hip -o /tmp/chain_trace.o
OBJS=$(ls *.o | grep -v '^chain.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/chain_trace.o -ldl -pthread -Wl,--version-script=libgpmi.map -o ../../tools/bin/libgpmi_chain_trace.so
echo built tools/bin/libgpmi_chain_trace.so
