#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "0,12288,0"; do
tag=${cfg//,/_}
GPMI_SUPER=$cfg timeout 400 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log" 2>&1
DB=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -name "*_results.db" | head -1)
python "$GRAFT_REPO_ROOT/tools/rocpd_mainstream.py" "$DB" 8 > "$GRAFT_REPO_ROOT/gpurun_out/mainstream_g_$tag.txt"
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag"
done
grep -A200 "^# window" "$GRAFT_REPO_ROOT/gpurun_out/mainstream_g_0_12288_0.txt" | head -170
