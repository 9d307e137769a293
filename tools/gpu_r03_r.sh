#!/bin/bash
# round 3, call R9: side launches capped at one per XCD only beside an update that runs as update256_kernel; threshold of the kernel
mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_twolevel.py -q -m gpu -x 2>&1 | grep -v amdgpu | tail -3
for m in 1024 512 2048; do
  GPMI_UPDATE256_MIN=$m timeout 300 python bench.py --n 50000 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('UPDATE256_MIN=$m N=50000', {k: round(j[k],2) for k in ('ms_per_step','fit_only_ms_per_step','predict_only_ms_per_step')}, 'frac', round(j['roofline']['frac'],4))"
done 2>&1 | tee $O/r_ab9.log
timeout 200 python bench.py --n 50000 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --mode sharded 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('blocked handle on one rank N=50000', round(j['ms_per_step'],2))" | tee -a $O/r_ab9.log
