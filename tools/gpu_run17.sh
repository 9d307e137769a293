#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof17" -- python "$GRAFT_REPO_ROOT/tools/fitc_bench.py" 1000000x4096 > "$GRAFT_REPO_ROOT/gpurun_out/fitc_c5.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof17 -name "*_results.db" | head -1)
python tools/rocpd_long.py "$DB" 3000 > gpurun_out/fitc_long.txt
tail -45 gpurun_out/fitc_long.txt
rm -rf gpurun_out/prof17
