#!/bin/bash
# gradient path bring-up: new tests first (bounded), then the whole GPU suite and a bench line
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gradient or optimize" -x > gpurun_out/grad.log 2>&1
echo "grad rc=$?" >> gpurun_out/grad.log
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "all rc=$?" >> gpurun_out/gpu_tests.log
timeout 300 python tools/grad_time.py > gpurun_out/grad_time.log 2>&1
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -5 gpurun_out/grad.log; tail -3 gpurun_out/gpu_tests.log; cat gpurun_out/grad_time.log; cat gpurun_out/bench.json
exit 0
