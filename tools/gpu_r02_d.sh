#!/bin/bash
# two-level factorisation with the super-block inverse: parity first, then the knob sweep
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_parity.py tests/test_gpu_fitc.py tests/test_reference_goldens.py -q -m gpu -x > gpurun_out/parity_d.log 2>&1; tail -15 gpurun_out/parity_d.log
N="50000" CFGS="6144,12288,24576 6144,12288,0 6144,10240,20480 4096,8192,16384 6144,12288,24576:4608:1024:16 6144,12288,0:3000" bash tools/super_sweep.sh 2>&1 | tee gpurun_out/super_sweep_d.log
GPMI_SUPER_INV=0 N="50000" CFGS="6144,12288,24576" bash tools/super_sweep.sh 2>&1 | tee -a gpurun_out/super_sweep_d.log
N="20000" CFGS="0,0,0:4608:256 6144,12288,0 6144,10240,0 4096,8192,16384 0,8192,0 6144,12288,0:3000" STEPS=8 bash tools/super_sweep.sh 2>&1 | tee -a gpurun_out/super_sweep_d.log
