#!/bin/bash
# round 3, call V3: the product build at the final tree: the suites that crashed under the experimental first-tile-from-the-queue change
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_parity.py -q -m gpu -x -k "lookahead or twolevel or width or pivot or posdef or fit_sizes or 256x128" 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/v3_tests.log
timeout 60 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -1
