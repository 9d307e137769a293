#!/bin/bash
# round 4, call G: cov kernels after the row-pair / early-exit changes (parity, then time); the overlap test; blocked one-rank block widths
mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fitc.py -q -m gpu -x -k "cov or limit or fit_sizes or all_kernels or tall or predict" > $O/r04_g_tests.log 2>&1; tail -2 $O/r04_g_tests.log
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -x -s -k "injected" > $O/r04_g_tests_dist.log 2>&1; grep -v amdgpu $O/r04_g_tests_dist.log | grep -E "passed|failed" | tail -2
cd /tmp; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for w in seard c3 f32d16; do
  P="$R/$O/pmc_cov_$w"; rm -rf "$P"; mkdir -p "$P"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -- python "$R/tools/cov_only.py" $w > "$P/kt.log" 2>&1
  grep -h "cov_\|scale_inputs" $(find "$P/kt" -name "*kernel_stats.csv") | cut -c1-70,170-260
  rm -rf "$P"
done
cd "$R"
echo "== blocked one rank, N = 20000: block widths"
for wd in 512 1024; do
GPMI_DIST_WD=$wd timeout 300 python bench.py --n 20000 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --mode sharded 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('n=20000 sharded WD=$wd: ms %.1f fit %.1f predict %.1f frac %.3f' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['predict_only_ms_per_step'], j['roofline']['frac']))"
done
