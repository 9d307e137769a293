#!/bin/bash
# two-level factorisation: parity first, then the knob sweep
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fitc.py tests/test_reference_goldens.py -q -m gpu -x > gpurun_out/parity_c.log 2>&1; tail -5 gpurun_out/parity_c.log
N="50000" CFGS="0,0,0 6144,0,0 6144,12288,0 6144,12288,24576 0,8192,0 8192,8192,0:4608:256" bash tools/super_sweep.sh 2>&1 | tee gpurun_out/super_sweep.log
N="20000" CFGS="0,0,0:4608:256 0,0,0 6144,0,0 6144,12288,0 0,8192,0" STEPS=8 bash tools/super_sweep.sh 2>&1 | tee -a gpurun_out/super_sweep.log
