#!/bin/bash
# sweep of the look-ahead threshold on the bench workload: CFGS="chol_min ..."
for cfg in ${CFGS:-"4608"}; do
  set -- ${cfg/:/ }
  GPMI_LOOKAHEAD_MIN=$1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('chol_min=$1', round(j['ms_per_step'],2), round(j['roofline']['achieved'],1), round(j['stage_ms_per_step']['predict'],2))"
done
