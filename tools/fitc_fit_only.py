"""Two FITC fits at BASELINE configs[4] (N = 1e6, M = 4096, SEArd) and nothing else: the workload of a `rocprofv3 --kernel-trace --stats` run whose
per-kernel totals, halved, are one gpmi_fitc_fit (tools/gpu_r06_v.sh -> profiles/r06_v_fitc_fit_kernel_stats.csv)."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
import numpy as np
import gpmi355x as g

n, m, d = 1000000, 4096, 8
rng = np.random.default_rng(20240501)
x = rng.uniform(size=(d, n)); xu = rng.uniform(size=(d, m))
y = np.sin(2.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
ll = [math.log(0.5) + 0.05 * k for k in range(d)]
gp = g.FITC(x, xu, y, g.MeanZero(), g.SEArd(ll, 0.0), math.log(0.1))
gp.set_params([v + 0.01 for v in gp.get_params()])
t = time.perf_counter(); gp.update_mll(); print(f"update_mll {time.perf_counter() - t:.3f} s, mll {gp.mll:.4f}")
