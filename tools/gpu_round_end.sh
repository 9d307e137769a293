#!/bin/bash
# everything the round's records need, in one box: full GPU suite, bench (with cpu_baseline), kernel-trace profile,
# PMC traffic passes, the other configs' timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests_full.log 2>&1; tail -3 gpurun_out/gpu_tests_full.log
timeout 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; cut -c1-600 gpurun_out/bench_full.json
timeout 600 python tools/scale_probe.py c3 f32_20k f32_100k > gpurun_out/scale_probe.log 2>&1; grep -v amdgpu gpurun_out/scale_probe.log
timeout 300 python tools/grad_time.py 2>&1 | grep -v amdgpu > gpurun_out/grad_time.log; cat gpurun_out/grad_time.log
timeout 300 python tools/fitc_bench.py 1000000x4096 2>&1 | grep -v amdgpu > gpurun_out/fitc_c5.log; cat gpurun_out/fitc_c5.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_end" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_end.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/prof_end -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" > gpurun_out/kernel_stats.csv
python tools/rocpd_groups.py "$DB" > gpurun_out/groups.txt
head -8 gpurun_out/kernel_stats.csv
rm -rf gpurun_out/prof_end
bash tools/gpu_pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1; tail -2 gpurun_out/pmc_bench.log
rm -rf gpurun_out/pmc_bench
