#!/bin/bash
# the bench-related records only: bench JSON, kernel-trace stats of the same command, PMC traffic passes
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; cut -c1-400 gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_end" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_end.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_end -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" > gpurun_out/kernel_stats.csv
python tools/rocpd_groups.py "$DB" > gpurun_out/groups.txt
head -4 gpurun_out/kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_end
bash tools/gpu_pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1; tail -1 gpurun_out/pmc_bench.log
rm -rf gpurun_out/pmc_bench
