#!/bin/bash
# full GPU suite (without the 200 s full-size file unless FULL=1) + bench + kernel-trace profile
mkdir -p gpurun_out
FILES="tests/test_gpu_parity.py tests/test_gpu_dist.py"
[ -n "$FULL" ] && FILES="tests"
timeout 1200 python -m pytest $FILES -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1
tail -4 gpurun_out/gpu_tests.log
timeout 300 python bench.py --steps 5 --warmup 2 ${CPU:---no-cpu-baseline} > gpurun_out/bench.json 2> gpurun_out/bench.err
cut -c1-1700 gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/gpurun_out/prof_tbp" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_tbp.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_tbp -name "*_results.db" | head -1)
python tools/rocpd_groups.py "$DB" > gpurun_out/groups.txt
python tools/rocpd_stats.py "$DB" > gpurun_out/kernel_stats.csv
head -24 gpurun_out/groups.txt
rm -rf gpurun_out/prof_tbp
