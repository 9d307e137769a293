import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import gpmi355x as g
from oracle import gp_oracle as G
from test_gpu_fitc import CASES, _case
for name, spec, n, d, m in CASES:
    x, xu, y, xs = _case(n, d, m, 31)
    ln = math.log(0.2)
    o64 = G.fitc_update_mll(spec, x, xu, y, ln, ("const", 0.25))
    old = G.fitc_update_mll_extended(spec, x, xu, y, ln, ("const", 0.25))
    gp = g.FITC(x, xu, y, g.MeanConst(0.25), g.from_spec(spec), ln)
    au = gp.cK.alpha_u()
    mu_r, S_r = G.fitc_predict_f(spec, xu, o64, xs, ("const", 0.25), full_cov=True)
    mu, var = gp.predict_f(xs)
    _, S = gp.predict_f(xs, full_cov=True)
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
    print(f"{name}: mll dev {gp.mll:.9f} o64 {o64['mll']:.9f} ext {old['mll']:.9f} | alpha vs ext {rel(gp.alpha, old['alpha']):.1e} vs o64 {rel(gp.alpha, o64['alpha']):.1e}"
          f" | alpha_u vs ext {rel(au, old['alpha_u']):.1e} vs o64 {rel(au, o64['alpha_u']):.1e} | mu vs o64 {rel(mu, mu_r):.1e} var vs o64 abs {np.abs(var - np.maximum(np.diag(S_r), 0)).max():.1e} S abs {np.abs(S - S_r).max():.1e}")
