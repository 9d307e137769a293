#!/bin/bash
# round 4, call H: small sweeps on the bench workload before the records run
mkdir -p gpurun_out; O=gpurun_out
for cfg in "GPMI_UPDATE256_MIN=1024" "GPMI_UPDATE256_MIN=512" "GPMI_UPDATE256_MIN=256" "GPMI_SUPER=8192,12288,20480" "GPMI_SUPER=8192,16384,28672"; do
  env $cfg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$cfg', 'n50000 ms %.1f fit %.1f frac %.3f' % (j['ms_per_step'], j['fit_only_ms_per_step'], j['roofline']['frac']))"
done 2>&1 | tee $O/r04_h_sweeps.log
