#!/bin/bash
# look-ahead threshold sweep on the bench workload
mkdir -p gpurun_out
for cfg in ${CFGS:-"0 0" "8 9472" "8 6144"}; do
  set -- ${cfg/:/ }
  GPMI_LOOKAHEAD=$1 GPMI_LOOKAHEAD_MIN=$2 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_la.json 2> gpurun_out/bench_la.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_la.json").read().strip().splitlines()[-1])
    print("L=$1 min=$2", "ms/step", round(j["ms_per_step"], 2), "syrk TF", round(j["roofline"]["achieved"], 2), {k: round(v, 1) for k, v in j["stage_ms_per_step"].items()})
except Exception as e:
    print("failed", e); print(open("gpurun_out/bench_la.err").read()[-1500:])
PY
done
