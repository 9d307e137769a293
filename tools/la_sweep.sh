#!/bin/bash
# sweep of the look-ahead knobs on the bench workload: CFGS="chol_min[:slots] ..."
for cfg in ${CFGS:-"4608"}; do
  set -- ${cfg/:/ }
  GPMI_LOOKAHEAD_MIN=$1 GPMI_LOOKAHEAD=${2:-8} timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('chol_min=$1 slots=${2:-8}', round(j['ms_per_step'],2), round(j['roofline']['achieved'],1), round(j['stage_ms_per_step']['predict'],2))"
done
