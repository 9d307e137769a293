#!/bin/bash
# round 3, call K: in-process device groups (gpmi_ctx_create with n_devices > 1) + the rest of the blocked suite
mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -q -m gpu -x --durations=5 > $O/k_tests.log 2>&1; grep -v amdgpu $O/k_tests.log | tail -14
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/k_group_timing.log
import math, sys, time
sys.path.insert(0, "gaussianprocesses.jl_amd"); sys.path.insert(0, ".")
import numpy as np
import gpmi355x as g
from gpmi355x import dist as gd
rng = np.random.default_rng(1)
n, d = 30000, 8
x = rng.uniform(size=(d, n)); y = np.sin(2 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n); xs = rng.uniform(size=(d, 512))
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
ref = g.GP(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1))
for devs in ([0], [0, 0], [0, 0, 0, 0]):
    ctx = g.Context(devices=devs)
    gp = gd.ShardedGPE(x, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1), ctx=ctx, block=1024)
    t0 = time.perf_counter(); gp.update_mll(); t1 = time.perf_counter(); mu, s2 = gp.predict_f(xs); t2 = time.perf_counter()
    print(f"group {devs}: update_mll {1e3*(t1-t0):.1f} ms predict {1e3*(t2-t1):.1f} ms mll rel {abs(gp.mll/ref.mll-1):.1e} (dense N={n})", flush=True)
    del gp; ctx.close()
PY
