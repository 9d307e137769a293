#!/bin/bash
# round 3, call A: CU-masked look-ahead A/B, new full-size parity tests, the widened bench line, critical paths
mkdir -p gpurun_out; O=gpurun_out
echo "== quick correctness with the CU-masked chain (default)"
timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -x > $O/a_tests_quick.log 2>&1; tail -3 $O/a_tests_quick.log
echo "== A/B: round-2 slots vs reserved CUs, look-ahead threshold sweep (N = 50000 + c2)"
for cfg in "GPMI_CUMASK=0" "GPMI_CUMASK=1" "GPMI_CUMASK=1 GPMI_LOOKAHEAD_MIN=3072" "GPMI_CUMASK=1 GPMI_LOOKAHEAD_MIN=2048" "GPMI_CUMASK=1 GPMI_LOOKAHEAD_MIN=3072 GPMI_SUPER=6144,10240,24576"; do
  env $cfg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$cfg', 'n50000 ms %.1f frac %.3f | c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']), j['c2']['stage_ms_per_step'])"
done 2>&1 | tee $O/a_ab.log
echo "== new full-size parity tests"
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fitc.py -q -m gpu -x -s -k "c3_n50000_composite_direct or n200000 or c5_full_size" > $O/a_tests_full.log 2>&1; grep -v amdgpu $O/a_tests_full.log | tail -12
echo "== full bench line"
timeout 900 python bench.py > $O/a_bench_full.json 2> $O/a_bench_full.err; echo "rc $?"; cut -c1-3000 $O/a_bench_full.json; tail -3 $O/a_bench_full.err
echo "== critical paths"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/prof_a" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_a.log" 2>&1
timeout 300 rocprofv3 --kernel-trace -d "$GRAFT_REPO_ROOT/$O/prof_a2" -- python "$GRAFT_REPO_ROOT/bench.py" --n 20000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$GRAFT_REPO_ROOT/$O/prof_a2.log" 2>&1
cd "$GRAFT_REPO_ROOT"
DB=$(find $O/prof_a -name "*_results.db" | head -1); python tools/rocpd_mainstream.py "$DB" > $O/a_bench_critical_path.txt; python tools/rocpd_stats.py "$DB" > $O/a_bench_kernel_stats.csv
DB=$(find $O/prof_a2 -name "*_results.db" | head -1); python tools/rocpd_mainstream.py "$DB" > $O/a_c2_critical_path.txt
head -12 $O/a_c2_critical_path.txt; tail -2 $O/a_c2_critical_path.txt; head -8 $O/a_bench_critical_path.txt
rm -rf $O/prof_a $O/prof_a2
