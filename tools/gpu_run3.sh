#!/bin/bash
set -u
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/peak.log 2>&1
import sys; sys.path.insert(0, "gaussianprocesses.jl_amd")
import gpmi355x as g
c = g.Context.default(0)
for bits in (64, 32):
    print("mfma peak", bits, [round(c.mfma_peak(bits), 2) for _ in range(3)], "TFLOP/s")
PY
cat gpurun_out/peak.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/pytest_parity.log 2>&1; echo "parity exit $?"; tail -n 15 gpurun_out/pytest_parity.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -n 2 gpurun_out/bench.log
