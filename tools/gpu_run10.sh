#!/bin/bash
# look-ahead Cholesky: reserved panel CUs sweep + correctness subset
mkdir -p gpurun_out
for R in 0 16 32 64; do
  GPMI_PANEL_CUS=$R timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_R$R.json 2> gpurun_out/bench_R$R.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_R$R.json").read().strip().splitlines()[-1])
    print("R=$R", "ms/step", round(j["ms_per_step"], 2), "syrk TF", round(j["roofline"]["achieved"], 2), j["stage_ms_per_step"])
except Exception as e:
    print("R=$R failed", e); print(open("gpurun_out/bench_R$R.err").read()[-2000:])
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "not gradient_all and not all_kernels" > gpurun_out/parity_subset.log 2>&1
tail -3 gpurun_out/parity_subset.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/fullsize.log 2>&1
tail -3 gpurun_out/fullsize.log
exit 0
