#!/bin/bash
set -u
mkdir -p gpurun_out
./tools/bin/mfma_bench > gpurun_out/mfma_bench.log 2>&1; cat gpurun_out/mfma_bench.log
timeout 600 python tools/yardstick.py > gpurun_out/yardstick.log 2>&1; cat gpurun_out/yardstick.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 1200 -p no:cacheprovider --durations=5 > gpurun_out/pytest_fullsize.log 2>&1; echo "fullsize exit $?"; tail -n 30 gpurun_out/pytest_fullsize.log
