#!/bin/bash
# round 4, call L: the blocked / sharded tests and the smoke at the final tree (the whole suite ran in call J before the last blocked.cpp changes)
mkdir -p gpurun_out; O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_abi.py -q -m "gpu or not gpu" -s > $O/r04_l_tests_dist.log 2>&1; grep -v amdgpu $O/r04_l_tests_dist.log | grep -E "passed|failed|^FAILED" | tail -4
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -1
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "f3_packed" > $O/r04_l_tests_f3.log 2>&1; tail -1 $O/r04_l_tests_f3.log
