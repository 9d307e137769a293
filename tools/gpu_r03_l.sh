#!/bin/bash
# round 3, call L: libgpmi without torch in the process (smoke: /opt/rocm's HIP runtime), and with torch imported first
mkdir -p gpurun_out; O=gpurun_out
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -3
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tail -8
import math, sys, time
sys.path.insert(0, "gaussianprocesses.jl_amd"); sys.path.insert(0, ".")
import numpy as np
import gpmi355x as g
assert "torch" not in sys.modules
rng = np.random.default_rng(1)
for n in (20000, 50000):
    d = 8
    x = rng.uniform(size=(d, n)); y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n); xs = rng.uniform(size=(d, 1024))
    ll = [math.log(0.5) + 0.05 * k for k in range(d)]
    gp = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1))
    ts = []
    for i in range(5):
        t0 = time.perf_counter(); gp.set_params([v + 1e-3 for v in gp.get_params()]); gp.update_mll(); gp.predict_f(xs); ts.append(time.perf_counter() - t0)
    print(f"no torch in the process (ROCm's own HIP runtime): N={n} step {1e3*min(ts):.1f} ms (min of 5), torch loaded: {'torch' in sys.modules}")
    del gp
ctx = g.Context(devices=[0, 0])
from gpmi355x import dist as gd
x = rng.uniform(size=(4, 3000)); y = np.sin(x.sum(axis=0)); 
gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd([0.0] * 4, 0.0), -1.0, ctx=ctx)
ref = g.GP(x, y, g.MeanZero(), g.SEArd([0.0] * 4, 0.0), -1.0)
print("device group without torch: mll rel", abs(gp.mll / ref.mll - 1), "torch loaded:", "torch" in sys.modules)
PY
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_twolevel.py -q -m gpu -x 2>&1 | grep -v amdgpu | tail -3
