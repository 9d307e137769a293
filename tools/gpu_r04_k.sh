#!/bin/bash
# round 4, call K: the reworked injected-latency test (N = 65 536, 5 ms), twice
mkdir -p gpurun_out; O=gpurun_out
for i in 1 2; do
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -x -s -k "injected" > $O/r04_k_tests_dist_$i.log 2>&1; grep -v amdgpu $O/r04_k_tests_dist_$i.log | grep -E "passed|failed|injected-latency" | cut -c1-1600 | tail -3
done
