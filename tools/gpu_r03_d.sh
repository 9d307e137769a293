#!/bin/bash
# round 3, call D: stream sets vs hardware queues (GPU_MAX_HW_QUEUES), virtual ranks after the test-double fix, cov multi-leaf kernel
mkdir -p gpurun_out; O=gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$*', 'n50000 ms %.1f frac %.3f | c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"; }
echo "== stream sets"
{ run GPMI_X=exclusive
  run GPMI_STREAM_SETS=both
  run GPMI_STREAM_SETS=both GPU_MAX_HW_QUEUES=8
  run GPU_MAX_HW_QUEUES=8
  env GPMI_X=c2alone timeout 300 python bench.py --n 20000 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('c2 alone (masked streams created first): ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac']))"
} 2>&1 | tee $O/d_streams.log
echo "== virtual ranks + cov kernels"
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py -q -m gpu -k "virtual or cov or fit_all_kernels or predict or noise" > $O/d_tests.log 2>&1; grep -v amdgpu $O/d_tests.log | tail -8
echo "== C3 composite cov"
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --secondary c3 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('n50000 ms %.1f cov %.2f ms %.0f GB/s | c3' % (j['ms_per_step'], j['stage_ms_per_step']['cov'], j['stage_ms_per_step']['cov_GBps']), json.dumps(j['c3']))" 2>&1 | tee $O/d_c3.log
