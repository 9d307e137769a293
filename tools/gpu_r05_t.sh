#!/bin/bash
# round 5, call T: where the 22 us of the one-wavefront potf2 go (phase marks inside diag64_body, trace build)
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/chain_trace.py --skip 3 --launches 1 50000 blocked --skip 0 --launches 1 20000 blocked > $O/r05_t_potf2_phases.log 2>&1
cat $O/r05_t_potf2_phases.log
