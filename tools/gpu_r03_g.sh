#!/bin/bash
# round 3, call G: dense path (mode per factorisation), blocked driver on one rank after the restructure, regression tests
mkdir -p gpurun_out; O=gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$*', 'n50000 ms %.1f frac %.3f | c2 ms %.2f frac %.3f' % (j['ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['roofline_frac']))"; }
{ run GPMI_X=default; } 2>&1 | tee $O/g_dense.log
echo "== blocked code path on one rank"
timeout 600 python bench.py --mode sharded --steps 5 --warmup 2 --no-cpu-baseline --secondary c2,c4 2> $O/g_sharded.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('sharded world 1: n50000 ms %.1f frac %.3f mll %.6f | c2 ms %.2f | c4 s %.3f mll %.4f' % (j['ms_per_step'], j['roofline']['frac'], j['config']['mll'], j['c2']['ms_per_step'], j['c4_single_gpu']['s_per_step'], j['c4_single_gpu']['mll']))" 2>&1 | tee $O/g_sharded.log
tail -2 $O/g_sharded.err | grep -v amdgpu
echo "== tests"
timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_twolevel.py tests/test_gpu_parity.py tests/test_gpu_fitc.py -q -m gpu -x -k "not c5_full_size and not m4096" > $O/g_tests.log 2>&1; grep -v amdgpu $O/g_tests.log | tail -5
