#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/dist_q.log 2>&1; tail -4 gpurun_out/dist_q.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k f3_packed --durations=3 > gpurun_out/f3_q.log 2>&1; tail -25 gpurun_out/f3_q.log
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
