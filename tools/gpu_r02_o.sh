#!/bin/bash
mkdir -p gpurun_out
for n in 20000 50000; do
for wd in 512 1024 2048; do
  GPMI_DIST_WD=$wd timeout 400 python bench.py --n $n --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --mode sharded 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['stage_ms_per_step']
print('n=$n sharded world=1 WD=$wd', 'ms', round(j['ms_per_step'],2), 'upd TF', round(j['roofline']['achieved'],1), 'upd ms', round(s['chol_trailing_update'],1), 'mll', repr(j['config']['mll']))"
done
done 2>&1 | tee gpurun_out/sharded_wd.log
