#!/bin/bash
# round 4, call B: functional confirmation of the round's first half on the device — hidden-visibility build, the AbstractPDMat surface on
# blocked handles, the all-gather straight into P, the injected-latency overlap test, C4 direct fp64 parity, the new bench parity fields
mkdir -p gpurun_out; O=gpurun_out
echo "== dist + abi tests"
timeout 1500 python -m pytest tests/test_abi.py tests/test_gpu_dist.py -q -m "gpu or not gpu" -x -s > $O/r04_b_tests_dist.log 2>&1; grep -v amdgpu $O/r04_b_tests_dist.log | grep -E "passed|failed|error|injected-latency" | tail -5
echo "== parity + two-level (quick regression)"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_twolevel.py -q -m gpu -x > $O/r04_b_tests_parity.log 2>&1; tail -2 $O/r04_b_tests_parity.log
echo "== C4 at its own size: fp32 dense / fp32 blocked vs fp64 packed"
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -s -k "n200000" > $O/r04_b_tests_c4.log 2>&1; grep -v amdgpu $O/r04_b_tests_c4.log | grep -E "C4|fp32|passed|failed|Error" | tail -6
echo "== bench line with the new parity fields"
timeout 1200 python bench.py --steps 3 --warmup 1 --secondary c4,grad,c5 > $O/r04_b_bench.json 2> $O/r04_b_bench.err; echo "rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_b_bench.json").readline())
print("ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"])
for k in ("c4_single_gpu", "grad", "c5"):
    print(k, json.dumps(j.get(k))[:900])
print("cpu", {k: j["cpu_baseline"].get(k) for k in ("value", "stage_s", "cov_numpy_vectorised_s", "fits_per_sec_with_numpy_cov")})
print("parity", j.get("parity"))
PY
tail -3 $O/r04_b_bench.err
