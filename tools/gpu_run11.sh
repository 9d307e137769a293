#!/bin/bash
mkdir -p gpurun_out
L=${1:-8}
cd /tmp && export TMPDIR=/tmp
GPMI_LOOKAHEAD=$L timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_la" -- python "$GRAFT_REPO_ROOT/tools/la_check.py" > "$GRAFT_REPO_ROOT/gpurun_out/prof_la.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_la -name "*_results.db" | head -1)
python tools/rocpd_timeline.py "$DB" 400 > gpurun_out/timeline_L$L.txt
tail -3 gpurun_out/prof_la.log; tail -2 gpurun_out/timeline_L$L.txt
rm -rf gpurun_out/prof_la
