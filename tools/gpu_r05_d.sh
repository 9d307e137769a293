#!/bin/bash
# round 5, call D: finer sweep of the super-panel thresholds x whole-CU switch at several sizes
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
for N in 8192 12288 20000 28000 50000; do echo "== N = $N"; timeout 900 python tools/knob_sweep.py $N fine 2>&1 | grep "^N="; done
} > $O/r05_d_sweeps.log 2>&1
cat $O/r05_d_sweeps.log
