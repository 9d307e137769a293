"""ONE fit of BASELINE configs[2] — N = 50 000, d = 8, (SEArd + Mat52Iso) + Noise, fp64 — for the PMC passes over its covariance
kernels (cov_multi_kernel: interior tiles; cov_kernel: the diagonal / edge tiles): tools/gpu_r03_records.sh."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gpmi355x as g
rng = np.random.default_rng(20240501)
n, d = 50000, 8
x = rng.uniform(size=(d, n)); y = np.sin(2 * np.pi * x).sum(axis=0) / d + 0.1 * rng.standard_normal(n)
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
spec = ("sum", ("sum", ("se_ard", ll, 0.0), ("mat52_iso", math.log(0.7), math.log(0.5))), ("noise", math.log(0.05)))
gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.1))
print("c3 mll", gp.mll)
