#!/bin/bash
# round 3, call W: N = 20000 on the plain stream with the 256 x 128 kernel for (nearly) every look-ahead update instead of reserved CUs + 128 x 128
mkdir -p gpurun_out
for cfg in "0 1" "0 200" "1 1024"; do
  set -- $cfg
  GPMI_CUMASK=$1 GPMI_UPDATE256_MIN=$2 timeout 100 python bench.py --n 20000 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('CUMASK=$1 UPDATE256_MIN=$2 N=20000', {k: round(j[k],2) for k in ('ms_per_step','fit_only_ms_per_step','predict_only_ms_per_step')}, 'frac', round(j['roofline']['frac'],4))"
done 2>&1 | tee gpurun_out/w_c2.log
