#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fitc.py -q -m gpu -x > gpurun_out/fitc_r.log 2>&1; tail -25 gpurun_out/fitc_r.log
