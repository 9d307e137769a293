#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_dist.log 2>&1; echo "dist exit $?"; tail -n 30 gpurun_out/pytest_dist.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest_parity.log 2>&1; echo "parity exit $?"; tail -n 5 gpurun_out/pytest_parity.log
