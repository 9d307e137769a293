#!/bin/bash
# round 5, call P: the task timeline of the persistent chain kernel from clock marks inside the kernel (tools/chain_trace.py, on the trace
# build tools/bin/libgpmi_chain_trace.so of tools/build_chain_trace.sh) — blocked one rank (the launch alone on the chip, 32 workgroups) at
# the two leading dimensions, and dense (beside the trailing update, 8 workgroups)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/chain_trace.py --skip 3 --launches 2 20000 blocked 50000 blocked --skip 2 --launches 3 20000 dense --skip 4 --launches 3 50000 dense > $O/r05_p_chain_trace.log 2>&1
cat $O/r05_p_chain_trace.log
