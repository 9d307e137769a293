#!/bin/bash
# round 3, call I: super-panel thresholds at N = 20000 with the fast (whole-CU) chain; trimmed C3 test
mkdir -p gpurun_out; O=gpurun_out
for sup in "8192,12288,24576" "8192,12288,16384" "6144,10240,16384" "4096,8192,16384" "6144,8192,14336"; do
  GPMI_SUPER=$sup timeout 200 python bench.py --n 20000 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('GPMI_SUPER=$sup  c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac']), {k: round(v, 2) for k, v in j['stage_ms_per_step'].items() if k != 'note'})"
done 2>&1 | tee $O/i_super_c2.log
for la in 2048 4096; do
  GPMI_LOOKAHEAD_MIN=$la timeout 200 python bench.py --n 20000 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('GPMI_LOOKAHEAD_MIN=$la  c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac']))"
done 2>&1 | tee -a $O/i_super_c2.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c3_n50000_composite_properties" --durations=3 2>&1 | grep -v amdgpu | tail -5
