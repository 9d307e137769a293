"""ONE fit whose covariance kernels the PMC passes of tools/gpu_r04_a.sh look at:
  cov_only.py c3       N = 50 000, d = 8, (SEArd + Mat52Iso) + Noise, fp64   (cov_multi_kernel interior tiles)
  cov_only.py seard    N = 50 000, d = 8, SEArd, fp64                         (cov_fast_kernel)
  cov_only.py f32d16   N = 50 000, d = 16, SEArd, fp32                        (cov_fast_kernel<float>, C4's element type and d)"""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gpmi355x as g
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
d = 16 if which == "f32d16" else 8
rng = np.random.default_rng(20240501)
x = rng.uniform(size=(d, n)); y = np.sin(2 * np.pi * x).sum(axis=0) / d + 0.1 * rng.standard_normal(n)
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
if which == "c3":
    spec = ("sum", ("sum", ("se_ard", ll, 0.0), ("mat52_iso", math.log(0.7), math.log(0.5))), ("noise", math.log(0.05)))
else:
    spec = ("se_ard", ll, 0.0)
gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.1), dtype=np.float32 if which == "f32d16" else np.float64)
print(which, "mll", gp.mll)
