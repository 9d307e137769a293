#!/bin/bash
# round 3, call T: the quick GPU suites + the >= 32768-row tests at the final tree (after the side-cap refinement)
mkdir -p gpurun_out; O=gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_twolevel.py tests/test_gpu_fullsize.py -q -m gpu -x -k "not n220000 and not composite_direct and not n200000 and not c2_n20000" --durations=5 2>&1 | grep -v amdgpu | tail -12 | tee $O/t_tests.log
timeout 120 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -1
