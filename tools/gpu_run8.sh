#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_all.log 2>&1; echo "tests exit $?"; tail -n 40 gpurun_out/pytest_all.log | cut -c1-250
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -n 1 gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof.log" 2>&1; echo "rocprof exit $?"
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" > gpurun_out/kernel_stats.csv
