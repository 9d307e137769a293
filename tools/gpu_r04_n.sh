#!/bin/bash
# round 4, call N: the whole -m gpu suite at the final tree, as the driver runs it (-x)
mkdir -p gpurun_out; O=gpurun_out
timeout 2700 python -m pytest tests -x -q -m gpu > $O/r04_n_tests.log 2>&1; grep -v amdgpu $O/r04_n_tests.log | tail -6
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -1
