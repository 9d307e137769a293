#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -n 1 gpurun_out/bench.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --mode sharded > gpurun_out/bench_sharded1.log 2>&1; echo "sharded(1) exit $?"; tail -n 3 gpurun_out/bench_sharded1.log | cut -c1-900
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --mode sharded > gpurun_out/bench_torchrun1.log 2>&1; echo "torchrun(1) exit $?"; tail -n 2 gpurun_out/bench_torchrun1.log | cut -c1-400
