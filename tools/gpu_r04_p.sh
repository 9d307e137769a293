#!/bin/bash
# round 4, call P: cov! split so that the first diagonal block is factored under its tail — parity, then the bench workloads
mkdir -p gpurun_out; O=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_parity.py tests/test_reference_goldens.py -q -m gpu -x > $O/r04_p_tests.log 2>&1; tail -2 $O/r04_p_tests.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "c2_n20000 or n20000_d16" > $O/r04_p_tests2.log 2>&1; tail -1 $O/r04_p_tests2.log
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2,c3 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('n50000 ms %.1f fit %.1f frac %.3f | c2 ms %.2f fit %.2f | c3 ms %.1f cov %.2f' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['roofline']['frac'], j['c2']['ms_per_step'], j['c2']['fit_only_ms_per_step'], j['c3']['ms_per_step'], j['c3']['cov_ms_per_step']))"
done | tee $O/r04_p_bench.log
