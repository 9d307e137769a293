"""GPU parity of the FITC path (gpmi_fitc_*) against the CPU oracle's restatement of
src/sparse/fully_indep_train_conditional.jl — SURVEY §8f rank 2.

Which oracle: ΣQR = Kuf Λ⁻¹ Kfu + Kuu has a condition number ~ n / (σ² · 1e-10); for smooth kernels its small pivots are
rounding noise in any fp64 Cholesky, so the fp64 (LAPACK) evaluation of the reference's statements is itself off by up to
1.4e-4 relative in mll on the SE cases below (tools/fitc_probe.py, LABBOOK.md §3.6).  mll / alpha / alpha_u are therefore
checked against the SAME statements evaluated in 80-bit arithmetic (oracle.fitc_update_mll_extended), where the device path
(which factors the well-conditioned whitened matrix B instead of ΣQR) agrees to ~1e-8; predictions are checked against the
fp64 oracle at the north-star's rtol 1e-5."""
import math

import numpy as np
import pytest

import gpmi355x as g
from oracle import gp_oracle as G

pytestmark = pytest.mark.gpu


def _case(n, d, m, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(d, n))
    xu = rng.uniform(size=(d, m))
    y = np.sin(3.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(d, 37))
    return x, xu, y, xs


CASES = [
    ("se_ard_d2", ("se_ard", [math.log(0.3), math.log(0.45)], 0.1), 1500, 2, 100),
    ("se_iso_d1", ("se_iso", math.log(0.2), 0.0), 1000, 1, 12),          # test_sparse.jl's shape: n = 1000, 12 inducing
    ("mat52+rq_d3", ("sum", ("mat52_iso", math.log(0.6), 0.1), ("rq_iso", 0.0, -0.5, 0.3)), 2100, 3, 300),
    ("prod_masked_d3", ("prod", ("masked", ("se_iso", math.log(0.5), 0.0), [0, 1]), ("mat32_iso", math.log(0.9), 0.2)), 900, 3, 65),
]


@pytest.mark.parametrize("name,spec,n,d,m", CASES, ids=[c[0] for c in CASES])
def test_fitc_fit_and_predict_match_the_oracle(name, spec, n, d, m):
    x, xu, y, xs = _case(n, d, m, 31)
    ln = math.log(0.2)
    ref = G.fitc_update_mll(spec, x, xu, y, ln, ("const", 0.25))
    ext = G.fitc_update_mll_extended(spec, x, xu, y, ln, ("const", 0.25))
    gp = g.FITC(x, xu, y, g.MeanConst(0.25), g.from_spec(spec), ln)
    assert abs(gp.mll - ext["mll"]) <= 1e-6 * abs(ext["mll"])
    assert abs(gp.mll - ref["mll"]) <= 1e-3 * abs(ref["mll"])           # the fp64 statement of the same thing
    np.testing.assert_allclose(gp.alpha, ext["alpha"], rtol=0, atol=1e-6 * np.abs(ext["alpha"]).max())
    au = gp.cK.alpha_u()
    np.testing.assert_allclose(au, ext["alpha_u"], rtol=0, atol=1e-3 * np.abs(ext["alpha_u"]).max())
    mu_r, S_r = G.fitc_predict_f(spec, xu, ref, xs, ("const", 0.25), full_cov=True)
    mu, var = gp.predict_f(xs)
    np.testing.assert_allclose(mu, mu_r, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(var, np.maximum(np.diag(S_r), 0.0), rtol=1e-5, atol=1e-7)
    mu2, S = gp.predict_f(xs, full_cov=True)
    np.testing.assert_allclose(mu2, mu_r, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(S, S_r, rtol=1e-5, atol=1e-7)
    muy, vy = gp.predict_y(xs)
    np.testing.assert_allclose(vy, var + math.exp(2 * ln), rtol=1e-12)


def test_fitc_parameter_updates_and_refit():
    """set_params! -> update_mll! on the same device handle (the optimiser's loop), test_sparse.jl:122-127."""
    spec = ("se_ard", [math.log(0.3), math.log(0.45)], 0.1)
    x, xu, y, xs = _case(1200, 2, 70, 33)
    gp = g.FITC(x, xu, y, g.MeanZero(), g.from_spec(spec), -1.0)
    first = gp.mll
    p = gp.get_params()
    q = [v + 0.1 for v in p]
    gp.set_params(q)
    gp.update_mll()
    spec2 = ("se_ard", [q[1], q[2]], q[3])
    ref = G.fitc_update_mll_extended(spec2, x, xu, y, q[0])
    assert abs(gp.mll - ref["mll"]) <= 1e-6 * abs(ref["mll"])
    gp.set_params(p)
    gp.update_mll()
    assert gp.mll == first  # deterministic: same inputs, same bits


def test_fitc_inducing_at_the_data_reproduces_the_exact_gp():
    spec = ("mat32_iso", math.log(0.5), 0.0)
    x, _, y, xs = _case(640, 2, 5, 35)
    sp = g.FITC(x, x, y, g.MeanZero(), g.from_spec(spec), math.log(0.3))
    ex = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.3))
    assert abs(sp.mll - ex.mll) < 1e-5 * abs(ex.mll)  # they differ by the two 1e-10 nuggets (5.5e-4 here, oracle alike)
    assert abs(sp.mll - G.fitc_update_mll(spec, x, x, y, math.log(0.3))["mll"]) < 1e-7 * abs(ex.mll)
    m1, v1 = sp.predict_f(xs)
    m2, v2 = ex.predict_f(xs)
    np.testing.assert_allclose(m1, m2, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v1, v2, rtol=1e-4, atol=1e-6)


def test_fitc_fp32():
    # a rough kernel and few inducing points: the 1e-10 nugget is below fp32's epsilon, so Kuu has to be
    # positive definite in fp32 on its own (the reference has no fp32 path at all)
    spec = ("mat12_iso", math.log(0.4), 0.0)
    x, xu, y, xs = _case(2000, 2, 20, 37)
    ref = G.fitc_update_mll(spec, x, xu, y, math.log(0.3))
    gp = g.FITC(x, xu, y, g.MeanZero(), g.from_spec(spec), math.log(0.3), dtype=np.float32)
    assert abs(gp.mll - ref["mll"]) <= 1e-2 * abs(ref["mll"])
    mu_r, v_r = G.fitc_predict_f(spec, xu, ref, xs)
    mu, var = gp.predict_f(xs)
    np.testing.assert_allclose(mu, mu_r, rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(var, v_r, rtol=1e-2, atol=1e-2)


def test_fitc_error_contract():
    x, xu, y, xs = _case(300, 2, 10, 39)
    k = g.SEArd([0.0, 0.0], 0.0)
    with pytest.raises(g.ArgumentError):
        g.FITC(x, xu[:1], y, g.MeanZero(), k, -1.0)          # inducing points of the wrong dimension
    gp = g.FITC(x, xu, y, g.MeanZero(), k, -1.0)
    with pytest.raises(g.ArgumentError):
        gp.predict_f(xs[:1])


GRAD_CASES = [  # (name, spec, d, m): Kuu conditioned 1e3 ... 6e4, where the fp64 literal restatement is good to 1e-9
    ("se_ard_d2_m12", ("se_ard", [math.log(0.3), math.log(0.45)], 0.1), 2, 12),
    ("mat52+se_d3_m40", ("sum", ("mat52_ard", [-0.6, -0.4, -0.5], 0.1), ("se_iso", -0.2, -0.4)), 3, 40),
    ("prod_rq_d3_m40", ("prod", ("rq_iso", -0.3, 0.1, 0.4), ("mat32_iso", 0.2, -0.1)), 3, 40),
    ("mat32_ard_d2_m60", ("mat32_ard", [-0.9, -0.7], 0.1), 2, 60),
    # composites whose leaves need the trace kernel's special cases: Noise (isapprox delta, only on the diagonal terms),
    # Masked + Const under a product, a FixedKernel that exposes three of six parameters
    ("se+noise_d3_m30", ("sum", ("se_ard", [-0.6, -0.4, -0.5], 0.1), ("noise", -1.5)), 3, 30),
    ("masked*const_d3_m30", ("prod", ("masked", ("mat52_iso", -0.4, 0.2), (0, 2)), ("const", -0.3)), 3, 30),
    ("fixed_d3_m30", ("fixed", ("sum", ("mat32_ard", [-0.6, -0.4, -0.5], 0.1), ("rq_iso", -0.2, -0.4, 0.3)), (0, 3, 5)), 3, 30),
]


@pytest.mark.parametrize("name,spec,d,m", GRAD_CASES, ids=[c[0] for c in GRAD_CASES])
def test_fitc_gradient_matches_the_oracle(name, spec, d, m):
    """update_dmll! on a FITC model: dmll_noise, dmll_mean!, dmll_kern! (fully_indep_train_conditional.jl:200-257 over
    subsetofregressors.jl:219-256) against the oracle's literal restatement, which tests/test_oracle.py checks by central
    differences of the FITC mll."""
    x, xu, y, _ = _case(1500, d, m, 41)
    ln = math.log(0.25)
    gp = g.FITC(x, xu, y, g.MeanConst(0.2), g.from_spec(spec), ln)
    gp.update_dmll()
    ref = G.fitc_update_dmll(spec, x, xu, y, ln, ("const", 0.2))
    scale = np.abs(ref["dmll"]).max()
    tol = 1e-5 if name.startswith("masked") else 1e-6   # cond(Kuu) = 2e6 there: the literal statement itself is good to 3e-7
    np.testing.assert_allclose(gp.dmll, ref["dmll"], rtol=tol, atol=0.1 * tol * scale)
    # the switches select the blocks (GPE.jl:298-324)
    gp.update_dmll(noise=False, domean=False)
    np.testing.assert_allclose(gp.dmll, ref["dkern"], rtol=tol, atol=0.1 * tol * scale)


def test_fitc_gradient_when_kuu_is_ill_conditioned():
    """40 inducing points under a smooth kernel: cond(Kuu) = 7e9, and the reference's statements evaluated literally in
    fp64 (LAPACK) are off by 5e-3 — like its mll (LABBOOK.md 3.6).  The device path works in the coordinates whitened by
    Luu and is checked against central differences of the 80-bit evaluation of the mll."""
    spec0 = [math.log(0.3), math.log(0.45), 0.1]
    x, xu, y, _ = _case(1500, 2, 40, 41)
    ln = math.log(0.25)
    gp = g.FITC(x, xu, y, g.MeanConst(0.2), g.SEArd(spec0[:2], spec0[2]), ln)
    gp.update_dmll(noise=False, domean=False)

    def f(th):
        return G.fitc_update_mll_extended(("se_ard", [th[0], th[1]], th[2]), x, xu, y, ln, ("const", 0.2))["mll"]

    h, fd = 1e-4, []
    for p in range(3):
        e = np.zeros(3)
        e[p] = h
        fd.append((f(np.array(spec0) + e) - f(np.array(spec0) - e)) / (2 * h))
    np.testing.assert_allclose(gp.dmll, fd, rtol=2e-3)
    lit = G.fitc_update_dmll(("se_ard", spec0[:2], spec0[2]), x, xu, y, ln, ("const", 0.2))["dkern"]
    assert np.abs(gp.dmll - fd).max() < np.abs(lit - fd).max()   # closer to the truth than the fp64 literal statement


def test_fitc_gradient_with_split_k_and_padding():
    """n large enough for the split-K form of W' diag(q) W (4 chunks), m not a multiple of 64."""
    spec = ("mat32_ard", [-0.9, -0.7], 0.1)
    x, xu, y, _ = _case(17000, 2, 70, 43)
    ln = math.log(0.3)
    gp = g.FITC(x, xu, y, g.MeanZero(), g.from_spec(spec), ln)
    gp.update_dmll()
    ref = G.fitc_update_dmll(spec, x, xu, y, ln)
    np.testing.assert_allclose(gp.dmll, ref["dmll"], rtol=1e-6, atol=1e-7 * np.abs(ref["dmll"]).max())


def test_optimize_runs_on_a_fitc_model():
    """optimize! (src/optimize.jl:19-37) with the device target and gradient of the FITC strategy: the mll goes up and the
    gradient at the end is small along the free directions."""
    spec = ("se_ard", [math.log(0.8), math.log(0.9)], 0.3)
    x, xu, y, _ = _case(2500, 2, 64, 45)
    gp = g.FITC(x, xu, y, g.MeanZero(), g.from_spec(spec), math.log(0.5))
    before = gp.mll
    res = g.optimize(gp, options={"maxiter": 25})
    assert gp.mll > before + 1.0 and np.isfinite(res.fun)
    gp.update_dmll()
    assert np.abs(gp.dmll).max() < 1e-2 * max(1.0, abs(gp.mll))


def test_fitc_large_n_woodbury_residual():
    """N = 131072, M = 1024, d = 8 (the split-K product runs with 16 chunks, the whitening over 2048 row blocks): alpha must
    satisfy (Kfu Kuu^-1 Kuf + Λ) alpha = y, checked matrix-free on the host from device-built covariances."""
    import scipy.linalg as sla
    n, m, d = 131072, 1024, 8
    rng = np.random.default_rng(20240501)
    x = rng.uniform(size=(d, n))
    xu = rng.uniform(size=(d, m))
    y = np.sin(2.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    k = g.SEArd([math.log(0.5) + 0.05 * j for j in range(d)], 0.0)
    gp = g.FITC(x, xu, y, g.MeanZero(), k, math.log(0.1))
    Kuf = np.asarray(g.cov(k, xu, x))
    Kuu = np.asarray(g.cov(k, xu)) + 1e-10 * np.eye(m)
    c = sla.cho_factor(Kuu)
    W = sla.solve_triangular(c[0], Kuf, trans="T", lower=False)
    lam = math.exp(2 * math.log(0.1)) + 1.0 - (W * W).sum(axis=0)
    a = np.asarray(gp.alpha, dtype=np.float64)
    res = W.T @ (W @ a) + lam * a - y
    assert np.abs(res).max() <= 1e-6 * np.abs(y).max()
    # determinant lemma on the host: logdet = logdet(I + W Λ^-1 W') + sum log Λ  (the two nuggets enter below 1e-6 here)
    B = np.eye(m) + (W / lam) @ W.T
    logdet = 2 * np.log(np.diag(sla.cholesky(B))).sum() + np.log(lam).sum()
    mll = -(y @ a + logdet + n * math.log(2 * math.pi)) / 2
    assert abs(gp.mll - mll) <= 1e-6 * abs(mll)


def test_fitc_c5_width_m4096_n262144_woodbury_residual():
    """BASELINE config 5's inducing width (M = 4096) at N = 262 144, d = 8, SEArd: the same matrix-free host checks as above
    (Woodbury residual of alpha, determinant lemma for the mll) with Kuf (8.6 GB) fetched from the device."""
    import scipy.linalg as sla
    n, m, d = 262144, 4096, 8
    rng = np.random.default_rng(20240502)
    x = rng.uniform(size=(d, n))
    xu = rng.uniform(size=(d, m))
    y = np.sin(2.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    k = g.SEArd([math.log(0.5) + 0.05 * j for j in range(d)], 0.0)
    gp = g.FITC(x, xu, y, g.MeanZero(), k, math.log(0.1))
    a = np.asarray(gp.alpha, dtype=np.float64)
    Kuu = np.asarray(g.cov(k, xu)) + 1e-10 * np.eye(m)
    c = sla.cholesky(Kuu, lower=False)
    W = np.asarray(g.cov(k, xu, x))                                  # Kuf, m x n
    W = sla.solve_triangular(c, W, trans="T", lower=False, overwrite_b=True)
    lam = math.exp(2 * math.log(0.1)) + 1.0 - np.einsum("ij,ij->j", W, W)
    assert np.all(lam > 0)
    res = W.T @ (W @ a) + lam * a - y
    assert np.abs(res).max() <= 1e-6 * np.abs(y).max()
    B = np.eye(m) + (W / lam) @ W.T
    logdet = 2 * np.log(np.diag(sla.cholesky(B))).sum() + np.log(lam).sum()
    mll = -(y @ a + logdet + n * math.log(2 * math.pi)) / 2
    assert abs(gp.mll - mll) <= 1e-6 * abs(mll)
    xs = rng.uniform(size=(d, 64))
    mu, var = gp.predict_f(xs)
    assert np.all(np.isfinite(mu)) and np.all(var >= 0) and np.all(var <= 1.0 + 1e-9)


def test_fitc_c5_full_size_n1e6_m4096_woodbury_and_determinant_lemma():
    """BASELINE configs[4] AT ITS OWN SIZE — N = 10^6, M = 4096, d = 8, SEArd, fp64 — checked through the size-independent
    properties of the reference's FITC algebra (fully_indep_train_conditional.jl:38-41, :80, :134-156):
        Woodbury residual   (W'W + Lambda) alpha = y - mu ,  W = Luu^-1 Kuf ,  Lambda = s2 + k(x,x) - diag(W'W)
        determinant lemma   logdet = logdet(I + W Lambda^-1 W') + sum log Lambda   ->  mll
    The 2 n m^2 flops of these checks run on the device as plain PyTorch fp64 (rocBLAS / rocSOLVER: an implementation that
    shares nothing with libgpmi), in column chunks so that nothing n x m is resident twice; torch's Kuf itself is pinned to
    the oracle's cov on a sample of entries."""
    import gc

    import torch

    gc.collect()
    torch.cuda.empty_cache()
    n, m, d = 1000000, 4096, 8
    rng = np.random.default_rng(20240503)
    x = rng.uniform(size=(d, n))
    xu = rng.uniform(size=(d, m))
    y = np.sin(2.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    ll = [math.log(0.5) + 0.05 * j for j in range(d)]
    ln = math.log(0.1)
    gp = g.FITC(x, xu, y, g.MeanZero(), g.SEArd(ll, 0.0), ln)
    a_host = np.asarray(gp.alpha, dtype=np.float64)
    mll_dev = gp.mll
    xs = rng.uniform(size=(d, 64))
    mu, var = gp.predict_f(xs)
    assert np.all(np.isfinite(mu)) and np.all(var >= 0) and np.all(var <= 1.0 + 1e-9)
    alpha_u = np.asarray(gp.cK.alpha_u(), dtype=np.float64) if hasattr(gp.cK, "alpha_u") else None
    del gp
    gc.collect()

    dev = torch.device("cuda", 0)
    w = torch.tensor(np.exp(-2.0 * np.asarray(ll)), dtype=torch.float64, device=dev)          # SEArd: il2 (se_ard.jl:31)
    xu_t = torch.tensor(xu.T.copy(), dtype=torch.float64, device=dev)                          # m x d
    a_t = torch.tensor(a_host, device=dev)
    y_t = torch.tensor(y, device=dev)

    def kmat(A, B):  # s2 exp(-r/2), r = sum_k il2_k (a_k - b_k)^2   (se_ard.jl:43), accumulated dimension by dimension
        r = torch.zeros((A.shape[0], B.shape[0]), dtype=torch.float64, device=dev)
        for k in range(d):
            diff = A[:, k:k + 1] - B[:, k][None, :]
            r.addcmul_(diff, diff, value=float(w[k]))
        return torch.exp_(r.mul_(-0.5))

    Kuu = kmat(xu_t, xu_t) + 1e-10 * torch.eye(m, dtype=torch.float64, device=dev)              # make_posdef! nugget
    Luu = torch.linalg.cholesky(Kuu)
    spec = ("se_ard", ll, 0.0)
    chunk = 65536
    B = torch.eye(m, dtype=torch.float64, device=dev)
    t = torch.zeros(m, dtype=torch.float64, device=dev)
    sum_log_lam = 0.0
    lam_all = torch.empty(n, dtype=torch.float64, device=dev)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        xc = torch.tensor(x[:, c0:c1].T.copy(), dtype=torch.float64, device=dev)
        Kuf = kmat(xu_t, xc)                                                                    # m x chunk
        if c0 == 0:   # torch's covariance entries against the oracle's
            np.testing.assert_allclose(Kuf[:48, :64].cpu().numpy(), G.cov(spec, xu[:, :48], x[:, :64]), rtol=1e-13, atol=1e-300)
        W = torch.linalg.solve_triangular(Luu, Kuf, upper=False)
        lam = math.exp(2 * ln) + 1.0 - (W * W).sum(dim=0)
        assert bool((lam > 0).all())
        lam_all[c0:c1] = lam
        sum_log_lam += float(torch.log(lam).sum())
        t += W @ a_t[c0:c1]
        B += (W / lam) @ W.T
        del Kuf, W, xc
    resid_max = 0.0
    for c0 in range(0, n, chunk):                                  # second pass: (W'W + Lambda) alpha - y, chunk by chunk
        c1 = min(n, c0 + chunk)
        xc = torch.tensor(x[:, c0:c1].T.copy(), dtype=torch.float64, device=dev)
        W = torch.linalg.solve_triangular(Luu, kmat(xu_t, xc), upper=False)
        res = W.T @ t + lam_all[c0:c1] * a_t[c0:c1] - y_t[c0:c1]
        resid_max = max(resid_max, float(res.abs().max()))
        del W, xc
    logdet = 2.0 * float(torch.log(torch.diagonal(torch.linalg.cholesky(B))).sum()) + sum_log_lam
    mll = -(float(y_t @ a_t) + logdet + n * math.log(2 * math.pi)) / 2
    print(f"[FITC C5 full size] mll device {mll_dev:.6f}, determinant lemma {mll:.6f} (rel {abs(mll_dev / mll - 1):.2e}); "
          f"Woodbury residual {resid_max:.2e} (|y| max {np.abs(y).max():.2f})")
    assert resid_max <= 1e-6 * np.abs(y).max()
    assert abs(mll_dev - mll) <= 1e-6 * abs(mll)


def test_fitc_tall_products_in_256x128_tiles_match_the_128_kernel(monkeypatch):
    """Round 4: FITC's n m^2 products (W = Kfu Luu^-T's updates, the split-K U'U'' batches, the gradient's overwriting rectangles) run
    on update256_kernel (rectangular / GEMM_OVERWRITE / batched forms) from 8192 rows on.  The same model through both kernels —
    fit, predict, gradient — agrees to rounding (n = 20 000, m = 1500: K = 1024 whitening updates, ragged edge tiles in both
    directions, 16-way split-K), and the result still satisfies the Woodbury identity."""
    n, m, d = 20000, 1500, 4
    rng = np.random.default_rng(5)
    x = rng.uniform(size=(d, n))
    xu = rng.uniform(size=(d, m))
    y = np.sin(2.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(d, 50))
    kern = lambda: g.SEArd([math.log(0.4) + 0.05 * j for j in range(d)], 0.1)
    out = []
    for u256 in ("1", "0"):
        monkeypatch.setenv("GPMI_UPDATE256", u256)
        gp = g.FITC(x, xu, y, g.MeanZero(), kern(), math.log(0.15), ctx=g.Context(0))
        mu, var = gp.predict_f(xs)
        gp.update_dmll()
        out.append((gp.mll, np.array(gp.alpha), mu, var, np.array(gp.dmll)))
        del gp
    a, b = out
    assert abs(a[0] - b[0]) <= 1e-10 * abs(b[0])
    np.testing.assert_allclose(a[1], b[1], rtol=0, atol=1e-8 * np.abs(b[1]).max())
    np.testing.assert_allclose(a[2], b[2], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a[3], b[3], rtol=1e-7, atol=1e-11)
    np.testing.assert_allclose(a[4], b[4], rtol=1e-7, atol=1e-8 * np.abs(b[4]).max())
