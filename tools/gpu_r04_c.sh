#!/bin/bash
# round 4, call C: the rewritten covariance kernels (parity first, then time and counters), CU partitions + injected-latency overlap test
mkdir -p gpurun_out; O=gpurun_out
echo "== cov parity (every kernel family, d sweep, composites, Noise) + fit parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_reference_goldens.py -q -m gpu -x > $O/r04_c_tests_parity.log 2>&1; tail -3 $O/r04_c_tests_parity.log
echo "== CU partitions + injected latency"
timeout 1200 python -m pytest tests/test_gpu_dist.py -q -m gpu -x -s -k "partitions or injected" > $O/r04_c_tests_dist.log 2>&1; grep -v amdgpu $O/r04_c_tests_dist.log | grep -E "passed|failed|injected-latency|CU partitions|Error|assert" | cut -c1-1500 | tail -8
echo "== cov kernels: time"
cd /tmp; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for w in seard c3 f32d16; do
  P="$R/$O/pmc_cov_$w"; rm -rf "$P"; mkdir -p "$P"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -- python "$R/tools/cov_only.py" $w > "$P/kt.log" 2>&1
  grep -h "cov_\|scale_inputs" $(find "$P/kt" -name "*kernel_stats.csv") | cut -c1-200 > "$R/$O/r04_c_cov_stats_$w.csv"; cat "$R/$O/r04_c_cov_stats_$w.csv"
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$P/sq1" -- python "$R/tools/cov_only.py" $w > "$P/sq1.log" 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$P/sq2" -- python "$R/tools/cov_only.py" $w > "$P/sq2.log" 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES WRITE_SIZE --output-format csv -d "$P/wr" -- python "$R/tools/cov_only.py" $w > "$P/wr.log" 2>&1
  python "$R/tools/pmc_cov_valu.py" "$P" 1250025000 > "$R/$O/r04_c_cov_pmc_$w.json" 2>&1
  rm -rf "$P"
done
cd "$R"
echo "== bench (quick) with c3"
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary c2,c3 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('n50000 ms %.1f frac %.3f stage' % (j['ms_per_step'], j['roofline']['frac']), j['stage_ms_per_step'])
print('c2 ms %.2f frac %.3f' % (j['c2']['ms_per_step'], j['c2']['roofline_frac']), j['c2']['stage_ms_per_step'])
print('c3', j['c3'])"
