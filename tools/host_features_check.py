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
# priors: target = mll + log prior, dtarget consistent with central differences of target, MAP differs from the ML estimate
gp2 = g.GP(x, y, g.MeanConst(0.0), g.SEIso(0.0, 0.0), math.log(0.3))
g.set_priors(gp2.kernel, [g.Normal(-1.0, 0.05), g.Normal(0.0, 1.0)])
g.set_priors(gp2.noise_param, [g.Normal(-1.0, 0.3)])
gp2.update_target_and_dtarget()
lp = g.prior_logpdf(gp2.kernel) + g.prior_logpdf(gp2.noise_param)
assert abs(gp2.target - (gp2.mll + lp)) < 1e-9 and lp < 0
th = np.array(gp2.get_params()); d = gp2.dtarget.copy()
for k in range(len(th)):
    h = 1e-5; tp = th.copy(); tm = th.copy(); tp[k] += h; tm[k] -= h
    gp2.set_params(tp); gp2.update_target(); fp = gp2.target
    gp2.set_params(tm); gp2.update_target(); fm = gp2.target
    assert abs((fp - fm) / (2 * h) - d[k]) < 1e-4 * max(1.0, abs(d[k])), (k, (fp - fm) / (2 * h), d[k])
gp2.set_params(th); gp2.update_target(); t0 = gp2.target
g.optimize(gp2, options={"maxiter": 60})
gp3 = g.GP(x, y, g.MeanConst(0.0), g.SEIso(0.0, 0.0), math.log(0.3))   # the same model without priors: ML estimate
g.optimize(gp3, options={"maxiter": 60})
ll_map, ll_ml = gp2.get_params()[2], gp3.get_params()[2]
print("log length-scale: MAP %.4f, ML %.4f (prior centred at -1.0, sd 0.05)" % (ll_map, ll_ml))
assert gp2.target > t0 and abs(ll_map + 1.0) < abs(ll_ml + 1.0)   # the prior moves the optimum towards its centre (test/optim.jl:37-52)
print("priors ok")
