#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_twolevel.py -q -m gpu -x > gpurun_out/dist_l.log 2>&1; tail -15 gpurun_out/dist_l.log
for n in 20000 50000; do
for mode in single sharded; do
  if [ $mode = sharded ]; then M="--mode sharded"; else M=""; fi
  timeout 400 python bench.py --n $n --steps 4 --warmup 1 --no-cpu-baseline --no-secondary $M 2>gpurun_out/bench_l_$mode_$n.err | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['stage_ms_per_step']
print('n=$n mode=$mode', 'ms', round(j['ms_per_step'],2), 'upd TF', round(j['roofline']['achieved'],1), 'upd ms', round(s['chol_trailing_update'],1), 'predict', round(s['predict'],2), 'mll', repr(j['config']['mll']), j['config']['parallelism'])"
done
done 2>&1 | tee gpurun_out/sharded_world1.log
tail -3 gpurun_out/bench_l_*.err 2>/dev/null | tail -12
