#!/bin/bash
# round 3, call Z: super-panel thresholds at N = 50000 now that the chain of a free-slot factorisation runs on whole CUs
mkdir -p gpurun_out
for sup in "8192,12288,16384" "8192,12288,20480"; do
  GPMI_SUPER=$sup timeout 60 python bench.py --n 50000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('SUPER=$sup N=50000', {k: round(j[k],2) for k in ('ms_per_step','fit_only_ms_per_step')}, 'frac', round(j['roofline']['frac'],4))"
done 2>&1 | tee gpurun_out/z_super.log
