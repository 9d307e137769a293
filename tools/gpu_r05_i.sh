#!/bin/bash
# round 5, call I: what does the instrumentation of bench.py's timed region cost at N = 20 000 now that it factors in the free-slot mode?
mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
{
for H in none all syrk; do
  echo "== bench.py --n 20000, events in the timed region: $H"
  GPMI_BENCH_PROFILE=$H timeout 300 python bench.py --n 20000 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('  ms/step %.2f fit %.2f predict %.2f  roofline %.3f launches %d' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['predict_only_ms_per_step'], j['roofline']['frac'], j['roofline']['launches']))"
done
echo "== default bench with c2 as a secondary object (after the N = 50 000 workload in the same context)"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary c2 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('  N=50000 ms/step %.1f roofline %.3f peak_measured %s' % (j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('peak_measured_samples'))); c=j['c2']; print('  c2 ms/step %.2f fit %.2f predict %.2f frac %.3f' % (c['ms_per_step'], c['fit_only_ms_per_step'], c['predict_only_ms_per_step'], c['roofline_frac']))"
for H in none all; do
  echo "== N = 50000, events: $H"
  GPMI_BENCH_PROFILE=$H timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('  ms/step %.2f fit %.2f predict %.2f' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['predict_only_ms_per_step']))"
done
} > $O/r05_i_instrumentation.log 2>&1
cat $O/r05_i_instrumentation.log
