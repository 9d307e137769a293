"""FITC at scale (BASELINE.json configs[4]: N = 1e6, M = 4096, SEArd, one MI355X): wall time of update_mll! and predict_f,
plus the Woodbury residual of alpha on a size the host can check."""
import math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gaussianprocesses.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gpmi355x as g

sizes = [(int(a), int(b)) for a, b in (s.split("x") for s in (sys.argv[1:] or ["131072x1024", "1000000x4096"]))]
d = 8
for n, m in sizes:
    rng = np.random.default_rng(20240501)
    x = rng.uniform(size=(d, n))
    xu = rng.uniform(size=(d, m))
    y = np.sin(2.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(d, 1024))
    ll = [math.log(0.5) + 0.05 * k for k in range(d)]
    k = g.SEArd(ll, 0.0)
    t0 = time.perf_counter()
    gp = g.FITC(x, xu, y, g.MeanZero(), k, math.log(0.1))
    t1 = time.perf_counter()
    gp.set_params([v + 0.01 for v in gp.get_params()])
    t2 = time.perf_counter(); gp.update_mll(); t3 = time.perf_counter()
    mu, var = gp.predict_f(xs); t4 = time.perf_counter()
    flops = 2.0 * n * m * m  # whitening n m^2 + SYRK n m^2
    t5 = time.perf_counter(); gp.update_dmll(); t6 = time.perf_counter()   # first call allocates two more n x m buffers
    t7 = time.perf_counter(); gp.update_dmll(); t8 = time.perf_counter()
    print(f"N={n} M={m}: update_dmll {t8 - t7:.3f} s (first call incl. allocation {t6 - t5:.3f} s), |dmll|_inf {np.abs(gp.dmll).max():.4e}", flush=True)
    print(f"N={n} M={m}: first fit incl. alloc/upload {t1 - t0:.3f} s, update_mll {t3 - t2:.3f} s ({flops / (t3 - t2) / 1e12:.1f} TFLOP/s on 2 n m^2), "
          f"predict_f(1024) {1e3 * (t4 - t3):.1f} ms, mll {gp.mll:.6f}", flush=True)
    if n * m <= 2 ** 28:  # host check of alpha = (Kfu Kuu^-1 Kuf + Lambda)^-1 r through matrix-free products
        import scipy.linalg as sla
        Kuf = np.asarray(g.cov(gp.kernel, xu, x))            # m x n
        Kuu = np.asarray(g.cov(gp.kernel, xu)) + 1e-10 * np.eye(m)
        c = sla.cho_factor(Kuu)
        W = sla.solve_triangular(c[0], Kuf, trans="T", lower=False)
        kdiag = math.exp(2.0 * gp.kernel.get_params()[-1])   # SEArd: sigma_f^2
        lam = math.exp(2 * gp.logNoise) + kdiag - (W * W).sum(axis=0)
        a = np.asarray(gp.alpha, dtype=np.float64)
        res = W.T @ (W @ a) + lam * a - y
        print(f"   Woodbury residual |Sigma alpha - r|_inf / |r|_inf = {np.abs(res).max() / np.abs(y).max():.2e}", flush=True)
