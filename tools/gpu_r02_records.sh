#!/bin/bash
# records: full GPU suite, the default bench line (with cpu_baseline and secondary objects), kernel stats, PMC passes
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/gpu_tests_full.log 2>&1; tail -14 gpurun_out/gpu_tests_full.log
timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; cut -c1-1200 gpurun_out/bench_full.json; tail -3 gpurun_out/bench_full.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_end" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/gpurun_out/prof_end.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_end -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" > gpurun_out/kernel_stats.csv
python tools/rocpd_groups.py "$DB" > gpurun_out/groups.txt
python tools/rocpd_mainstream.py "$DB" > gpurun_out/mainstream.txt
head -8 gpurun_out/kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_end
bash tools/gpu_pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1; tail -2 gpurun_out/pmc_bench.log
rm -rf gpurun_out/pmc_bench
timeout 600 python tools/fitc_bench.py 1000000x4096 2>&1 | grep -v amdgpu > gpurun_out/fitc_c5.log; cat gpurun_out/fitc_c5.log
