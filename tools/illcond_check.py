"""Accuracy of the blocked device Cholesky on a covariance whose only regularisation is a 1e-10 nugget
(exact GP on FITC's inducing points with sigma^2 = 1e-10): logdet against LAPACK and against an 80-bit reference."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gpmi355x as g
from oracle import gp_oracle as G
LD = np.longdouble
def chol_ld(A):
    A = A.astype(LD).copy(); n = A.shape[0]; L = np.zeros_like(A)
    for j in range(n):
        L[j, j] = np.sqrt(A[j, j] - np.dot(L[j, :j], L[j, :j]))
        if j + 1 < n:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L
for (m, d, ll) in ((100, 2, (0.3, 0.45)), (300, 2, (0.3, 0.45)), (12, 1, (0.2,))):
    rng = np.random.default_rng(31)
    xu = rng.uniform(size=(d, m)); y = rng.standard_normal(m)
    spec = ("se_ard", [math.log(v) for v in ll], 0.1)
    K = G.cov(spec, xu); K[np.diag_indices_from(K)] += 1e-10
    truth = 2 * float(np.log(np.diag(chol_ld(K))).sum())
    lap = 2 * float(np.log(np.diag(np.linalg.cholesky(K))).sum())
    try:
        gp = g.GP(xu, y, g.MeanZero(), g.from_spec(spec), math.log(1e-5))
        dev = gp.cK.logdet()
    except Exception as e:
        dev = float("nan"); print("device failed:", e)
    print(f"m={m} d={d} refine={os.environ.get('GPMI_REFINE')}: logdet truth {truth:.6f}  lapack err {lap - truth:+.2e}  device err {dev - truth:+.2e}")
