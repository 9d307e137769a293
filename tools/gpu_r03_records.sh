#!/bin/bash
# round 3 records: full GPU suite, the default bench line (cpu_baseline + parity + secondary objects), kernel stats of the bench command,
# the PMC passes over the bench command (HBM traffic / MFMA busy of the trailing update) and over one C3 fit (cov kernels)
mkdir -p gpurun_out; O=gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --durations=10 > $O/r_gpu_tests_full.log 2>&1; grep -v amdgpu $O/r_gpu_tests_full.log | tail -16
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -2 | tee $O/r_smoke.log
timeout 900 python bench.py > $O/r_bench_full.json 2> $O/r_bench_full.err; echo "bench rc $?"; cut -c1-900 $O/r_bench_full.json; tail -3 $O/r_bench_full.err | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_end" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_end.log" 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find $O/prof_end -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" > $O/r_kernel_stats.csv
python tools/rocpd_groups.py "$DB" > $O/r_groups.txt
python tools/rocpd_mainstream.py "$DB" > $O/r_mainstream.txt 2>&1
head -8 $O/r_kernel_stats.csv | cut -c1-200; head -5 $O/r_mainstream.txt
rm -rf $O/prof_end
bash tools/gpu_pmc_bench.sh > $O/r_pmc_bench.log 2>&1; tail -2 $O/r_pmc_bench.log | cut -c1-1200
rm -rf $O/pmc_bench
# (the C3 covariance PMC record, profiles/r03_c3_cov_pmc.json, is from the previous records run: cov.hip has not changed since)
