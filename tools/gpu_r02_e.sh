#!/bin/bash
# where does the main stream's time go in the two-level factorisation? kernel trace of one config + critical-path breakdown
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "6144,12288,24576" "6144,12288,0"; do
tag=${cfg//,/_}
GPMI_SUPER=$cfg timeout 400 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log" 2>&1
DB=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -name "*_results.db" | head -1)
python "$GRAFT_REPO_ROOT/tools/rocpd_mainstream.py" "$DB" | tee "$GRAFT_REPO_ROOT/gpurun_out/mainstream_$tag.txt"
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag"
done
cd "$GRAFT_REPO_ROOT"
N="50000" CFGS="6144,12288,24576:4608:1024:16 6144,12288,24576:3000 6144,12288,24576:8000" bash tools/super_sweep.sh 2>&1 | tee gpurun_out/super_sweep_e.log
N="20000" CFGS="0,0,0:4608:256 6144,12288,0:3000 6144,12288,0:8000 6144,12288,0:4608:1024:16" STEPS=8 bash tools/super_sweep.sh 2>&1 | tee -a gpurun_out/super_sweep_e.log
