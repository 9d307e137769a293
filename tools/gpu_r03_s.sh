#!/bin/bash
# round 3, call S: ablation of update256_kernel (tools build)
mkdir -p gpurun_out
timeout 300 python tools/update256_ablate.py 2>&1 | grep -v amdgpu | tee gpurun_out/s_ablate.log
