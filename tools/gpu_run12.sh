#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/parity.log 2>&1
tail -5 gpurun_out/parity.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json | cut -c1-1800
