#!/bin/bash
# round 3, call Y: N = 20000 with the masked update a few workgroup slots short of full (the dispatcher ignores the mask when it deals workgroups to shader engines)
mkdir -p gpurun_out
for sl in 0 16 32 8; do
  GPMI_MASKED_SLACK=$sl timeout 60 python bench.py --n 20000 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('MASKED_SLACK=$sl N=20000', {k: round(j[k],2) for k in ('ms_per_step','fit_only_ms_per_step')}, 'frac', round(j['roofline']['frac'],4))"
done 2>&1 | tee gpurun_out/y_slack.log
