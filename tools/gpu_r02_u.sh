#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/fitc_bench.py 1000000x4096 2>&1 | grep -v amdgpu | tee gpurun_out/fitc_c5.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_c2" -- python "$GRAFT_REPO_ROOT/bench.py" --n 20000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/gpurun_out/prof_c2.log" 2>&1
DB=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_c2" -name "*_results.db" | head -1)
python "$GRAFT_REPO_ROOT/tools/rocpd_mainstream.py" "$DB" > "$GRAFT_REPO_ROOT/gpurun_out/mainstream_c2.txt"
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_c2"
head -34 "$GRAFT_REPO_ROOT/gpurun_out/mainstream_c2.txt"
