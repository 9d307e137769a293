#!/bin/bash
mkdir -p gpurun_out
N="20000" CFGS="8192,12288,24576:4608 8192,12288,24576:7000 8192,12288,24576:9000 8192,12288,24576:11000 8192,12288,24576:14000 10240,14336,24576:11000 6144,12288,24576:9000" STEPS=8 bash tools/super_sweep.sh 2>&1 | tee gpurun_out/la_sweep_v.log
N="50000" CFGS="8192,12288,24576:9000 8192,12288,24576:11000" STEPS=4 bash tools/super_sweep.sh 2>&1 | tee -a gpurun_out/la_sweep_v.log
