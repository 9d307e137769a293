#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_groups" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_groups.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_groups -name "*_results.db" | head -1)
python tools/rocpd_groups.py "$DB" > gpurun_out/groups.txt
python tools/rocpd_stats.py "$DB" > gpurun_out/kernel_stats.csv
cat gpurun_out/groups.txt
rm -rf gpurun_out/prof_groups
