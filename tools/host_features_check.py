"""Composite means and bound-constrained optimize! through the real device object (small N), against the oracle."""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianprocesses.jl_amd")); sys.path.insert(0, ROOT)
import gpmi355x as g
from oracle import gp_oracle as G
rng = np.random.default_rng(3)
d, n = 2, 300
x = rng.uniform(0, 3, (d, n)); y = np.sin(2 * x[0]) + 0.3 * x[1] ** 2 + 0.1 * rng.standard_normal(n)
m = g.MeanConst(0.2) + g.MeanPoly(np.array([[0.1, 0.05], [0.0, 0.3]])) * g.MeanConst(1.0)
spec = ("se_ard", [0.0, 0.2], 0.1)
gp = g.GP(x, y, m, g.from_spec(spec), math.log(0.2))
ref = G.update_mll(spec, x, y - m.mean(x), math.log(0.2))
print("mll device %.9f oracle %.9f" % (gp.mll, ref["mll"]))
assert abs(gp.mll - ref["mll"]) < 1e-7 * abs(ref["mll"])
res = g.optimize(gp, options={"maxiter": 15}, noisebounds=([-3.0], [-1.0]), kernbounds=([-1.0] * 3, [0.5] * 3))
p = gp.get_params()
print("optimised params", np.round(p, 4), "mll", round(gp.mll, 6))
assert -3.0 - 1e-9 <= p[0] <= -1.0 + 1e-9 and all(-1.0 - 1e-9 <= v <= 0.5 + 1e-9 for v in p[-3:])
assert gp.mll > ref["mll"]
print("ok")
