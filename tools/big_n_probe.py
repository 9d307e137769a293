"""Largest exact fp64 fit that fits one MI355X comfortably (one N x N buffer, no second copy, no distance caches):
N = 150 000, d = 8, SEArd — 180 GB for the factor."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocesses.jl_amd"))
import numpy as np
import gpmi355x as g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
d, p = 8, 256
rng = np.random.default_rng(20240501)
x = rng.uniform(size=(d, n)); y = np.sin(2 * np.pi * x).sum(axis=0) / d + 0.1 * rng.standard_normal(n); xs = rng.uniform(size=(d, p))
t0 = time.perf_counter()
gp = g.GP(x, y, g.MeanZero(), g.SEArd([math.log(0.5) + 0.05 * k for k in range(d)], 0.0), math.log(0.1))
t1 = time.perf_counter()
mu, s2 = gp.predict_f(xs)
t2 = time.perf_counter()
print(f"N={n} fp64: GP() incl. allocation + upload + fit {t1 - t0:.1f} s ({n**3 / 3 / (t1 - t0) / 1e12:.1f} TFLOP/s Cholesky-equivalent incl. everything), "
      f"predict_f({p}) {t2 - t1:.2f} s, mll {gp.mll:.6g}, finite {bool(np.isfinite(mu).all() and np.isfinite(s2).all() and (s2 >= 0).all())}")
