#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_fitc.py -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof16" -- python "$GRAFT_REPO_ROOT/tools/fitc_bench.py" 1000000x4096 > "$GRAFT_REPO_ROOT/gpurun_out/fitc_c5.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof16 -name "*_results.db" | head -1)
python tools/rocpd_groups.py "$DB" > gpurun_out/fitc_groups.txt
grep "N=" gpurun_out/fitc_c5.log; head -16 gpurun_out/fitc_groups.txt
rm -rf gpurun_out/prof16
