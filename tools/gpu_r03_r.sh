#!/bin/bash
# round 3, call R8: the update256 tests in both look-ahead modes; N = 20000 with free slots + update256 instead of reserved CUs
mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_twolevel.py -q -m gpu -x -k "256x128" 2>&1 | grep -v amdgpu | tail -3
for cfg in "1 1" "0 1" "0 0"; do
  set -- $cfg
  GPMI_CUMASK=$1 GPMI_UPDATE256=$2 timeout 300 python bench.py --n 20000 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('CUMASK=$1 UPDATE256=$2 N=20000', {k: round(j[k],2) for k in ('ms_per_step','fit_only_ms_per_step','predict_only_ms_per_step')}, 'frac', round(j['roofline']['frac'],4))"
done 2>&1 | tee $O/r_ab8.log
