#!/bin/bash
# round 3, call U: the driver's invocation run by hand at the final tree
mkdir -p gpurun_out
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/u_bench_driver_like.json 2> gpurun_out/u_bench_driver_like.err; echo "rc $?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/u_bench_driver_like.json"))
print({k: j[k] for k in ("value", "ms_per_step", "fit_only_ms_per_step", "predict_only_ms_per_step", "steps", "warmup")}, "frac", j["roofline"]["frac"], "parity", j["parity"]["ok"], "c2", j["c2"]["ms_per_step"], "c4", j["c4_single_gpu"]["s_per_step"], "c5", j["c5"].get("update_mll_s"))
PY
