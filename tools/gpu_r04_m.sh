#!/bin/bash
# round 4, call M: the atomic epilogue of update256_kernel (GPMI_UPDATE256_ATOMIC=1) against the load / add / store one: isolated, in the
# factorisation, and through the tests that compare the kernel with the oracle
mkdir -p gpurun_out; O=gpurun_out
for a in 0 1; do echo "== GPMI_UPDATE256_ATOMIC=$a isolated"; GPMI_UPDATE256_ATOMIC=$a timeout 300 python tools/update256_vs_128.py f64 2>&1 | grep -v amdgpu | head -4; done | tee $O/r04_m_atomic.log
GPMI_UPDATE256_ATOMIC=1 timeout 300 python tools/update256_vs_128.py f32 2>&1 | grep -v amdgpu | head -3 | tee -a $O/r04_m_atomic.log
for a in 0 1; do
GPMI_UPDATE256_ATOMIC=$a timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('atomic=$a n50000 ms %.1f fit %.1f frac %.3f mll %.6f' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['roofline']['frac'], j['config']['mll']))"
done 2>&1 | tee -a $O/r04_m_atomic.log
GPMI_UPDATE256_ATOMIC=1 timeout 900 python -m pytest tests/test_gpu_twolevel.py -q -m gpu -x -k "256x128" 2>&1 | tail -2 | tee -a $O/r04_m_atomic.log
