#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twolevel.py -q -m gpu -x > gpurun_out/parity_f.log 2>&1; tail -3 gpurun_out/parity_f.log
cd /tmp && export TMPDIR=/tmp
for cfg in "6144,12288,24576"; do
tag=${cfg//,/_}
GPMI_SUPER=$cfg timeout 400 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log" 2>&1
DB=$(find "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -name "*_results.db" | head -1)
python "$GRAFT_REPO_ROOT/tools/rocpd_mainstream.py" "$DB" | tee "$GRAFT_REPO_ROOT/gpurun_out/mainstream_f_$tag.txt"
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag"
done
cd "$GRAFT_REPO_ROOT"
N="50000" CFGS="6144,12288,24576 6144,12288,0 6144,10240,16384 4096,8192,16384 8192,12288,24576" bash tools/super_sweep.sh 2>&1 | tee gpurun_out/super_sweep_f.log
N="20000" CFGS="0,0,0:4608:256 6144,12288,0 6144,10240,0 4096,8192,16384 0,8192,0 8192,0,0" STEPS=8 bash tools/super_sweep.sh 2>&1 | tee -a gpurun_out/super_sweep_f.log
