#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/parity_j.log 2>&1; tail -3 gpurun_out/parity_j.log
N="50000 20000" CFGS="8192,12288,24576" bash tools/super_sweep.sh 2>&1 | tee gpurun_out/super_sweep_j.log
GPMI_WHITEN_INV=0 N="50000 20000" CFGS="8192,12288,24576" bash tools/super_sweep.sh 2>&1 | tee -a gpurun_out/super_sweep_j.log
