#!/bin/bash
# round 3, call M: the single-GPU bench without torch in the process (ROCm's own HIP runtime)
mkdir -p gpurun_out; O=gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2,c3,grad 2> $O/m_bench.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('n50000 ms %.1f frac %.3f | c2 ms %.2f frac %.3f | c3 ms %.1f cov %.2f | grad %.3f s' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac'], j['c3']['ms_per_step'], j['c3']['cov_ms_per_step'], j['grad']['s_per_call']))
print(json.dumps(j['stage_ms_per_step']))" 2>&1 | tee $O/m_bench.log
tail -2 $O/m_bench.err | grep -v amdgpu
GPMI_CUMASK=0 timeout 300 python bench.py --n 20000 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('GPMI_CUMASK=0 c2 alone ms %.2f' % j['ms_per_step'])" | tee -a $O/m_bench.log
timeout 300 python bench.py --n 20000 --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('default      c2 alone ms %.2f' % j['ms_per_step'])" | tee -a $O/m_bench.log
timeout 600 python -m pytest tests/test_gpu_dist.py -q -m gpu -x 2>&1 | grep "passed\|failed" | tail -2
