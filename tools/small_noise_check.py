"""Exact path with very small noise: device mll against LAPACK (oracle) and the 80-bit value, with and without the rows64
refinement step (GPMI_REFINE)."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gpmi355x as g
from oracle import gp_oracle as G
rng = np.random.default_rng(5)
n, d = 1500, 2
x = rng.uniform(size=(d, n)); y = np.sin(4 * x.sum(axis=0)) + 0.05 * rng.standard_normal(n)
spec = ("se_ard", [math.log(0.3), math.log(0.4)], 0.0)
for ln in (-2.0, -4.0, -6.0, -8.0, -10.0):
    try:
        ref = G.update_mll(spec, x, y, ln)["mll"]
    except Exception as e:
        ref = float("nan")
    K = G.cov(spec, x); K[np.diag_indices_from(K)] += math.exp(2 * ln)
    try:
        L = G._chol_lower_ld(K)
        z = np.zeros(n, dtype=np.longdouble)
        for i in range(n):
            z[i] = (y[i] - L[i, :i] @ z[:i]) / L[i, i]
        ext = float(-(z @ z + 2 * np.log(np.diag(L)).sum() + n * np.longdouble(math.log(2 * math.pi))) / 2)
    except Exception as e:
        ext = float("nan")
    try:
        dev = g.GP(x, y, g.MeanZero(), g.from_spec(spec), ln).mll
    except Exception as e:
        dev = float("nan"); msg = str(e)[:60]
    print(f"refine={os.environ.get('GPMI_REFINE')} logNoise={ln}: 80-bit {ext:.6f}  lapack err {ref - ext:+.2e}  device err {dev - ext:+.2e}", flush=True)
