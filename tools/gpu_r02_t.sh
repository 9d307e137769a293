#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_twolevel.py tests/test_gpu_fitc.py tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/t_tests.log 2>&1; tail -4 gpurun_out/t_tests.log
for n in 20000 50000; do
for pf in 1 0; do
  GPMI_POTRF256=$pf timeout 300 python bench.py --n $n --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['stage_ms_per_step']
print('n=$n potrf256=$pf', 'ms', round(j['ms_per_step'],2), 'upd TF', round(j['roofline']['achieved'],1), 'upd ms', round(s['chol_trailing_update'],1), 'predict', round(s['predict'],2), 'mll', repr(j['config']['mll']))"
done
done 2>&1 | tee gpurun_out/potrf256.log
for pf in 1 0; do
  GPMI_POTRF256=$pf timeout 300 python bench.py --n 20000 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --mode sharded 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['stage_ms_per_step']
print('n=20000 sharded world=1 potrf256=$pf', 'ms', round(j['ms_per_step'],2))"
done 2>&1 | tee -a gpurun_out/potrf256.log
