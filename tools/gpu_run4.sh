#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest_parity.log 2>&1; echo "parity exit $?"; tail -n 12 gpurun_out/pytest_parity.log
python tools/gemm_ablate.py > gpurun_out/gemm_ablate3.log 2>&1; cat gpurun_out/gemm_ablate3.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -n 1 gpurun_out/bench.log
