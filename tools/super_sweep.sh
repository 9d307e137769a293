#!/bin/bash
# sweep of the two-level factorisation knobs on the bench workload.
#   CFGS="min512,min1024,min2048[:lookahead_min[:whiten_super[:slots]]] ..."   N="50000 20000"
for n in ${N:-50000}; do
for cfg in ${CFGS:-"6144,16384,0"}; do
  IFS=: read -r sup lam ws slots <<< "$cfg"
  env GPMI_SUPER=$sup ${lam:+GPMI_LOOKAHEAD_MIN=$lam} ${ws:+GPMI_WHITEN_SUPER=$ws} ${slots:+GPMI_LOOKAHEAD=$slots} \
  timeout 300 python bench.py --n $n --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read()); s=j['stage_ms_per_step']
print('n=$n super=$sup la_min=${lam:-def} whiten=${ws:-def} slots=${slots:-def} inv=${GPMI_SUPER_INV:-1}', 'ms', round(j['ms_per_step'],2), 'upd TF', round(j['roofline']['achieved'],1), 'upd ms', round(s['chol_trailing_update'],1), 'panel ms', round(s['panel_potf2_trsm_update'],1), 'predict', round(s['predict'],2), 'mll', repr(j['config']['mll']))"
done
done
