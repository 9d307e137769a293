#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/dist_m.log 2>&1; tail -3 gpurun_out/dist_m.log
for n in 50000; do
for sm in own default host; do
  GPMI_DIST_STREAM=$sm timeout 400 python bench.py --n $n --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --mode sharded 2>gpurun_out/bench_m_$sm.err | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['stage_ms_per_step']
print('n=$n stream=$sm', 'ms', round(j['ms_per_step'],2), 'upd TF', round(j['roofline']['achieved'],1), 'upd ms', round(s['chol_trailing_update'],1), 'mll', repr(j['config']['mll']))"
done
done 2>&1 | tee gpurun_out/sharded_streams.log
