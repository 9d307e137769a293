#!/bin/bash
# round 3, call J: split-K for the predict V products
mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_twolevel.py tests/test_gpu_fullsize.py -q -m gpu -x -k "predict or two_level or whiten or loo or gradient_and_loo or c2_n20000 or every_width" 2>&1 | grep -v amdgpu | tail -4
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --secondary c2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('n50000 ms %.1f predict %.2f | c2 ms %.2f predict %.2f' % (j['ms_per_step'], j['stage_ms_per_step']['predict'], j['c2']['ms_per_step'], j['c2']['stage_ms_per_step']['predict']))" 2>&1 | tee $O/j_predict.log
