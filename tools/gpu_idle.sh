#!/bin/bash
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_idle" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_idle.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_idle -name "*_results.db" | head -1)
python tools/rocpd_idle.py "$DB"; rm -rf gpurun_out/prof_idle
