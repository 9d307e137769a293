#!/bin/bash
# round 2: parity tests, the N = 50 000 bench line, kernel-trace stats of that command (PMC passes are run separately)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/gpu_tests_full.log 2>&1; tail -25 gpurun_out/gpu_tests_full.log
grep -h "^\[fp32\|^\[N=" gpurun_out/gpu_tests_full.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; cut -c1-3000 gpurun_out/bench_full.json; tail -3 gpurun_out/bench_full.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_end" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/gpurun_out/prof_end.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_end -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" > gpurun_out/kernel_stats.csv
python tools/rocpd_groups.py "$DB" > gpurun_out/groups.txt
head -14 gpurun_out/kernel_stats.csv
cat gpurun_out/groups.txt | head -30
rm -rf gpurun_out/prof_end
