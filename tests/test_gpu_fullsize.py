"""Parity at BASELINE.json's full sizes.

C2 (N=20000, d=8, SEArd, fp64) and, since round 3, C3 (N=50000, d=8, (SEArd+Mat52Iso)+Noise, fp64) are compared DIRECTLY with
the CPU oracle (host LAPACK: about a minute resp. two).  C3 is additionally — and C4 (N=200000, fp32), which has no fp64
counterpart that fits anywhere, exclusively — checked through size-independent properties of the result on sparse probes:
    residual      (K + s2 I) alpha = y - mu           (K rows rebuilt by the ORACLE for ~1000 random rows)
    factor        |L^-1 (K v)|^2 = v' K v             for a v supported on ~100 points  (pins the whole factor)
    mll           -(y'alpha + logdet + n log 2pi)/2   recomputed from the parts
    predict       sigma2 in [0, k(x,x)], mu finite; at training inputs mu ~ y (test/gp.jl:47-50)
"""
import math
import os

import numpy as np
import pytest

import gpmi355x as g
from oracle import gp_oracle as G

pytestmark = pytest.mark.gpu

LL8 = [math.log(0.5) + 0.05 * k for k in range(8)]


import time


class _Laps:
    """stage clock of the three long tests: printed, and appended to gpurun_out/test_laps.log when that directory exists (where the suite's minutes go)"""

    def __init__(self, name):
        self.name, self.t0, self.laps = name, time.perf_counter(), []

    def __call__(self, what):
        t = time.perf_counter()
        self.laps.append((what, t - self.t0))
        self.t0 = t

    def done(self):
        line = f"[laps] {self.name}: " + ", ".join(f"{w} {s:.1f}s" for w, s in self.laps)
        print(line)
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "test_laps.log"), "a") as fh:
                fh.write(line + "\n")


def _assemble_upper_parallel(spec, x, log_noise):
    """update_cK! of the oracle (oracle/cov_oracle.c: the reference-order scalar loops) for the UPPER triangle dpotrf('U') reads, column
    blocks on the host's cores (ctypes releases the GIL): the single-threaded whole-matrix call was half a minute of this file at N = 50 000."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import c_oracle

    n = x.shape[1]
    K = np.empty((n, n), order="F")
    w = 1000

    def block(j0):
        j1 = min(n, j0 + w)
        K[:j1, j0:j1] = c_oracle.cov(spec, x[:, :j1], x[:, j0:j1])
        K[j1:, j0:j1] = 0.0

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        list(ex.map(block, range(0, n, w)))
    K[np.diag_indices(n)] += math.exp(2.0 * log_noise)
    return K


def _Kv_chunked(spec, x, noise_var, V, chunk=2500):
    """(K + noise_var I) V with K rebuilt by the oracle, a block of rows at a time."""
    n = x.shape[1]
    out = np.empty_like(V)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        out[s:e] = G.cov(spec, x[:, s:e], x) @ V + noise_var * V[s:e]
    return out


def test_c2_n20000_direct_vs_oracle():
    x, y, xs = G.synthetic_inputs(20000, 8, p=1024)
    spec = ("se_ard", LL8, 0.0)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.1))
    mu, s2 = gp.predict_f(xs)
    ref = G.update_mll(spec, x, y, math.log(0.1))
    mu_o, s2_o = G.predict_f(spec, x, ref, xs)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-9)            # north_star bar: 1e-5
    np.testing.assert_allclose(gp.alpha, ref["alpha"], rtol=1e-5, atol=1e-6 * np.abs(ref["alpha"]).max())
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(s2, s2_o, rtol=1e-5, atol=1e-9)


def test_c3_n50000_composite_properties():
    """Size-independent properties on sparse probes the oracle can rebuild in seconds (the DIRECT comparison against the oracle's
    full factorisation is test_c3_n50000_composite_direct_vs_oracle below)."""
    n = 50000
    x, y, xs = G.synthetic_inputs(n, 8, p=512)
    spec = ("sum", ("sum", ("se_ard", LL8, 0.0), ("mat52_iso", math.log(0.7), math.log(0.5))), ("noise", math.log(0.05)))
    log_noise = math.log(0.1)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), log_noise)
    nv = math.exp(2 * log_noise)
    rng = np.random.default_rng(0)
    # residual of the solve on 1024 random rows (K rows rebuilt by the oracle)
    rows = np.sort(rng.choice(n, 1024, replace=False))
    r = G.cov(spec, x[:, rows], x) @ gp.alpha + nv * gp.alpha[rows] - y[rows]
    assert np.abs(r).max() <= 1e-8 * max(1.0, np.abs(y).max())
    # the factor: |L^-1 K v|^2 == v'Kv for a v supported on 96 points (K v needs 96 columns of K only)
    S = np.sort(rng.choice(n, 96, replace=False))
    vS = rng.standard_normal(96)
    Kv = G.cov(spec, x, x[:, S]) @ vS
    Kv[S] += nv * vS
    w = gp.cK.whiten(Kv)
    assert float(w @ w) == pytest.approx(float(vS @ Kv[S]), rel=1e-10)
    # logdet, independently of the device's reduction: 2 Σ log U_ii summed on the host from the factor's diagonal ...
    dg = np.asarray(gp.cK.factor_diag(), dtype=np.float64)
    assert np.all(dg > 0)
    logdet_host = 2.0 * float(np.sum(np.log(dg)))
    assert gp.cK.logdet() == pytest.approx(logdet_host, rel=1e-12)
    # ... and the diagonal itself against LAPACK on a leading block: the first m pivots of the factor are the Cholesky of
    # the leading m x m block of K + s2 I, so their partial log-determinant must be LAPACK's
    m = 4096
    Kb = G.cov(spec, x[:, :m]) + nv * np.eye(m)
    Ub = np.linalg.cholesky(Kb)
    np.testing.assert_allclose(dg[:m], np.diag(Ub), rtol=1e-9)
    assert 2.0 * float(np.sum(np.log(dg[:m]))) == pytest.approx(2.0 * float(np.sum(np.log(np.diag(Ub)))), rel=1e-11)
    # mll assembled from its parts, with the host-side logdet
    assert gp.mll == pytest.approx(-(float(y @ gp.alpha) + logdet_host + G.LOG2PI * n) / 2, rel=1e-12)
    # predict
    mu, s2 = gp.predict_f(xs)
    kdiag = 1.0 + 0.25 + 0.05 ** 2
    assert np.all(np.isfinite(mu)) and np.all(s2 >= 0) and np.all(s2 <= kdiag * (1 + 1e-12))
    mu_t, s2_t = gp.predict_y(x[:, :256])
    np.testing.assert_allclose(mu_t, y[:256], atol=0.5)
    # a second, independent route to mu: K*' alpha with K* from the oracle
    np.testing.assert_allclose(mu, G.cov(spec, xs, x) @ gp.alpha, rtol=1e-7, atol=1e-9)


# --------------------------------------------------------------------------------------------
# C4's precision and dimension (fp32, d = 16, SEArd) where it matters: at sizes where fp32 round-off has accumulated.
# The reference has no fp32 path; north_star's bar is rtol 1e-2 against the fp64 result.
# --------------------------------------------------------------------------------------------
LL16 = [math.log(0.5) + 0.05 * k for k in range(16)]


def test_c4_fp32_n20000_d16_vs_fp64_oracle():
    x, y, xs = G.synthetic_inputs(20000, 16, p=256)
    spec = ("se_ard", LL16, 0.0)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.1), dtype=np.float32)
    mu, s2 = gp.predict_f(xs)
    ref = G.update_mll(spec, x, y, math.log(0.1))
    mu_o, s2_o = G.predict_f(spec, x, ref, xs)
    print(f"[fp32 N=20000 d=16] mll {gp.mll:.4f} vs fp64 oracle {ref['mll']:.4f} (rel {abs(gp.mll / ref['mll'] - 1):.2e}); "
          f"max|dmu| {np.abs(mu - mu_o).max():.2e}, max|ds2| {np.abs(s2 - s2_o).max():.2e}")
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-2)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-2, atol=1e-2 * np.abs(mu_o).max())
    np.testing.assert_allclose(s2, s2_o, rtol=1e-2, atol=1e-2 * np.abs(s2_o).max())
    np.testing.assert_allclose(gp.alpha, ref["alpha"], rtol=0, atol=1e-2 * np.abs(ref["alpha"]).max())


def test_c4_fp32_n100000_d16_properties_and_fp64_device():
    """N = 100 000, d = 16, fp32 (40 GB factor): positive definite, solve residual, factor identity and mll-from-parts on
    sparse probes the oracle can rebuild, and agreement with the fp64 device path (80 GB) at rtol 1e-2."""
    n, d = 100000, 16
    x, y, xs = G.synthetic_inputs(n, d, p=256)
    spec = ("se_ard", LL16, 0.0)
    log_noise = math.log(0.1)
    nv = math.exp(2 * log_noise)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), log_noise, dtype=np.float32)   # raises PosDefException if not PD
    a = np.asarray(gp.alpha, dtype=np.float64)
    rng = np.random.default_rng(5)
    # residual of (K + s2 I) alpha = y on 512 random rows (K rows rebuilt by the oracle in fp64)
    rows = np.sort(rng.choice(n, 512, replace=False))
    r = G.cov(spec, x[:, rows], x) @ a + nv * a[rows] - y[rows]
    print(f"[fp32 N=100000] residual max {np.abs(r).max():.3e} (|y| max {np.abs(y).max():.2f})")
    assert np.abs(r).max() <= 1e-2 * np.abs(y).max()
    # the factor: |L^-1 (K v)|^2 == v' K v for a v supported on 64 points (K v needs 64 columns of K only)
    S = np.sort(rng.choice(n, 64, replace=False))
    vS = rng.standard_normal(64)
    KS = G.cov(spec, x, x[:, S])                       # n x 64
    Kv = KS @ vS
    Kv[S] += nv * vS
    w = np.asarray(gp.cK.whiten(Kv.astype(np.float32)), dtype=np.float64)
    assert float(w @ w) == pytest.approx(float(vS @ Kv[S]), rel=1e-2)
    # logdet from the fetched diagonal, mll from its parts
    dg = np.asarray(gp.cK.factor_diag(), dtype=np.float64)
    assert np.all(dg > 0)
    logdet_host = 2.0 * float(np.sum(np.log(dg)))
    assert gp.cK.logdet() == pytest.approx(logdet_host, rel=1e-6)
    assert gp.mll == pytest.approx(-(float(y @ a) + logdet_host + G.LOG2PI * n) / 2, rel=1e-5)
    mu, s2 = gp.predict_f(xs)
    assert np.all(np.isfinite(mu)) and np.all(s2 >= 0) and np.all(s2 <= 1.0 + 1e-5)
    np.testing.assert_allclose(mu, G.cov(spec, xs, x) @ a, rtol=1e-2, atol=1e-3)
    mll32, alpha32 = gp.mll, a
    del gp
    # the same model in fp64 on the device (itself checked against the oracle at N = 20 000 and by properties at 50 000)
    gp64 = g.GP(x, y, g.MeanZero(), g.from_spec(spec), log_noise)
    mu64, s264 = gp64.predict_f(xs)
    print(f"[N=100000 d=16] mll fp32 {mll32:.3f} fp64 {gp64.mll:.3f} (rel {abs(mll32 / gp64.mll - 1):.2e}); "
          f"max|dmu| {np.abs(mu - mu64).max():.2e}, max|ds2| {np.abs(s2 - s264).max():.2e}")
    r64 = G.cov(spec, x[:, rows], x) @ gp64.alpha + nv * gp64.alpha[rows] - y[rows]
    assert np.abs(r64).max() <= 1e-8 * np.abs(y).max()
    assert mll32 == pytest.approx(gp64.mll, rel=1e-2)
    np.testing.assert_allclose(mu, mu64, rtol=1e-2, atol=1e-2 * np.abs(mu64).max())
    np.testing.assert_allclose(s2, s264, rtol=1e-2, atol=1e-2 * np.abs(s264).max())
    np.testing.assert_allclose(alpha32, gp64.alpha, rtol=0, atol=1e-2 * np.abs(gp64.alpha).max())


def test_f3_packed_n220000_fp64_on_one_device_block_diagonal():
    """N = 220 000 in fp64 on ONE device — past the ~180 000 ceiling of a full N x N buffer — in packed
    storage (stripes of block-rows that stop at their own diagonal, csrc/blocked.cpp behind gpmi_gp_create_blocked).  Size-independent exactness:
    220 clusters of 1000 points, 100 length-scale units apart, make K + s2 I EXACTLY block diagonal (exp(-r^2 / 2 l^2)
    underflows to 0.0 across clusters), so mll, alpha and the predictions of the full factorisation — which does all
    N^3 / 3 flops, knows nothing of the zeros, and whose 1024-row blocks and 8192-row stripes straddle the 1000-point
    clusters — must equal those of 220 independent 1000-point GPs, which the oracle computes in seconds."""
    import gc

    import torch

    gc.collect()                  # 205 GB: everything earlier tests left behind has to be gone
    torch.cuda.empty_cache()
    nc, m, d = int(os.environ.get("GPMI_F3_CLUSTERS", "220")), 1000, 2
    n = nc * m
    rng = np.random.default_rng(42)
    x = rng.uniform(0.0, 1.0, size=(d, n))
    x[0] += 100.0 * np.repeat(np.arange(nc), m)
    y = np.sin(6.0 * (x[0] % 100.0)) + x[1] + 0.1 * rng.standard_normal(n)
    spec = ("se_iso", math.log(0.3), 0.0)
    ln = math.log(0.1)
    lap = _Laps(f"f3 packed N={n}")
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), ln, packed=True)
    lap("device fit (packed)")
    assert gp.nobs == n and gp.cK.nstripes > 20
    print(f"[f3] N = {n}: packed factor {gp.cK.factor_bytes / 1e9:.1f} GB in {gp.cK.nstripes} stripes, mll {gp.mll:.6f}")
    packed_bytes = gp.cK.factor_bytes
    assert packed_bytes < 0.56 * 8.0 * n * n, packed_bytes          # the full square would be 387 GB: it does not fit
    # reference: the clusters one by one
    mll_ref, alpha_ref = 0.0, np.empty(n)
    refs = {}
    for c in range(nc):
        sl = slice(c * m, (c + 1) * m)
        ref = G.update_mll(spec, x[:, sl], y[sl], ln)
        mll_ref += ref["mll"]
        alpha_ref[sl] = ref["alpha"]
        if c in (0, 57, nc - 1):
            refs[c] = ref
    lap("oracle cluster fits")
    assert gp.mll == pytest.approx(mll_ref, rel=1e-9)
    np.testing.assert_allclose(gp.alpha, alpha_ref, rtol=1e-6, atol=1e-8 * np.abs(alpha_ref).max())
    # predictions next to three clusters (first stripe, a middle one, the last block)
    xs = np.concatenate([rng.uniform(0.0, 1.0, size=(d, 16)) + np.array([[100.0 * c], [0.0]]) for c in refs], axis=1)
    mu, s2 = gp.predict_f(xs)
    for j, c in enumerate(refs):
        sl = slice(c * m, (c + 1) * m)
        mu_o, s2_o = G.predict_f(spec, x[:, sl], refs[c], xs[:, 16 * j:16 * (j + 1)])
        np.testing.assert_allclose(mu[16 * j:16 * (j + 1)], mu_o, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(s2[16 * j:16 * (j + 1)], s2_o, rtol=1e-5, atol=1e-9)
    lap("predict + checks")
    del gp
    _free_device_memory()
    lap("free")
    lap.done()


# --------------------------------------------------------------------------------------------
# Round 3: every BASELINE config parity-checked AT ITS OWN SIZE.
# --------------------------------------------------------------------------------------------
def _free_device_memory():
    import gc

    import torch

    gc.collect()
    torch.cuda.empty_cache()


def test_c3_n50000_composite_direct_vs_oracle():
    """BASELINE configs[2] — N = 50 000, d = 8, (SEArd + Mat52Iso) + Noise, fp64 — DIRECTLY against the CPU oracle: cov! +
    nugget by the reference-order C loop (oracle/cov_oracle.c), LAPACK dpotrf / dpotrs on the host (the reference's
    make_posdef! and `cK \\ y`), predict_f as src/GP.jl:64-79.  About two minutes of host time; bar: north_star's 1e-5."""
    import scipy.linalg as sla

    from oracle import c_oracle

    _free_device_memory()
    lap = _Laps("c3 direct N=50000")
    n, p = 50000, 256
    x, y, xs = G.synthetic_inputs(n, 8, p=p)
    spec = ("sum", ("sum", ("se_ard", LL8, 0.0), ("mat52_iso", math.log(0.7), math.log(0.5))), ("noise", math.log(0.05)))
    log_noise = math.log(0.1)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), log_noise)
    mu, s2 = gp.predict_f(xs)
    mll_dev, alpha_dev = gp.mll, np.asarray(gp.alpha, dtype=np.float64)
    del gp
    lap("device fit + predict")
    K = _assemble_upper_parallel(spec, x, log_noise)                        # update_cK!: cov! + exp(2 logNoise) on the diagonal
    lap("oracle cov!")
    U, info = sla.lapack.dpotrf(K, lower=0, clean=0, overwrite_a=1)         # make_posdef!
    lap("host dpotrf")
    del K
    assert info == 0
    alpha = sla.cho_solve((U, False), y)
    mll = -(float(y @ alpha) + 2.0 * float(np.sum(np.log(np.diag(U)))) + G.LOG2PI * n) / 2.0   # GPE.jl:210
    Kc = c_oracle.cov(spec, x, xs)
    mu_o = Kc.T @ alpha
    Lck = sla.solve_triangular(U, Kc, trans="T", lower=False, overwrite_b=True)
    kdiag = 1.0 + 0.25 + 0.05 ** 2
    s2_o = np.maximum(kdiag - np.sum(Lck * Lck, axis=0), 0.0)
    lap("host solves")
    lap.done()
    print(f"[C3 direct] mll device {mll_dev:.6f} oracle {mll:.6f} (rel {abs(mll_dev / mll - 1):.2e}); "
          f"max|dmu| {np.abs(mu - mu_o).max():.2e}, max|ds2| {np.abs(s2 - s2_o).max():.2e}")
    assert mll_dev == pytest.approx(mll, rel=1e-9)
    np.testing.assert_allclose(alpha_dev, alpha, rtol=1e-5, atol=1e-6 * np.abs(alpha).max())
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(s2, s2_o, rtol=1e-5, atol=1e-9)


def test_c4_fp32_n200000_d16_properties_at_full_size():
    """BASELINE configs[3]'s size — N = 200 000, d = 16, SEArd, fp32 (160 GB factor on ONE device).  Size-independent properties
    evaluated by the fp64 oracle on sparse probes — the solve residual (K + s2 I) alpha = y on 512 rows, the factor identity
    |L^-1 K v|^2 = v'Kv, the leading 4096 pivots against LAPACK in fp64, logdet from the fetched diagonal, mll from its parts,
    mu = K*' alpha — and (round 4) mll / alpha / mu / s2 DIRECTLY against the fp64 fit of the same model in packed storage
    (~180 GB), for the dense fp32 handle and for the blocked fp32 handle; bar 1e-2 (fp32)."""
    _free_device_memory()
    lap = _Laps("c4 N=200000")
    n, d = 200000, 16
    x, y, xs = G.synthetic_inputs(n, d, p=256)
    spec = ("se_ard", LL16, 0.0)
    log_noise = math.log(0.1)
    nv = math.exp(2 * log_noise)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), log_noise, dtype=np.float32)   # raises PosDefException if not PD
    lap("fp32 dense fit")
    a = np.asarray(gp.alpha, dtype=np.float64)
    rng = np.random.default_rng(7)
    rows = np.sort(rng.choice(n, 512, replace=False))
    Krows = G.cov(spec, x[:, rows], x)                 # (kept: the fp64 fit's residual below reads the same 512 rows)
    r = Krows @ a + nv * a[rows] - y[rows]
    lap("oracle rows + residual")
    print(f"[fp32 N=200000] mll {gp.mll:.3f}; residual max {np.abs(r).max():.3e} (|y| max {np.abs(y).max():.2f})")
    assert np.abs(r).max() <= 1e-2 * np.abs(y).max()
    S = np.sort(rng.choice(n, 64, replace=False))
    vS = rng.standard_normal(64)
    KS = G.cov(spec, x, x[:, S])
    Kv = KS @ vS
    Kv[S] += nv * vS
    w = np.asarray(gp.cK.whiten(Kv.astype(np.float32)), dtype=np.float64)
    assert float(w @ w) == pytest.approx(float(vS @ Kv[S]), rel=1e-2)
    dg = np.asarray(gp.cK.factor_diag(), dtype=np.float64)
    assert np.all(dg > 0)
    m = 4096
    Ub = np.linalg.cholesky(G.cov(spec, x[:, :m]) + nv * np.eye(m))
    np.testing.assert_allclose(dg[:m], np.diag(Ub), rtol=1e-3)
    logdet_host = 2.0 * float(np.sum(np.log(dg)))
    assert gp.cK.logdet() == pytest.approx(logdet_host, rel=1e-6)
    assert gp.mll == pytest.approx(-(float(y @ a) + logdet_host + G.LOG2PI * n) / 2, rel=1e-5)
    mu, s2 = gp.predict_f(xs)
    assert np.all(np.isfinite(mu)) and np.all(s2 >= 0) and np.all(s2 <= 1.0 + 1e-5)
    np.testing.assert_allclose(mu, G.cov(spec, xs, x) @ a, rtol=1e-2, atol=1e-3)
    # predictions at training inputs reproduce the data within the noise (test/gp.jl:47-50)
    mu_t, _ = gp.predict_f(x[:, :128])
    np.testing.assert_allclose(mu_t, y[:128], atol=0.5)
    mll32, mu32, s232 = gp.mll, np.array(mu), np.array(s2)
    lap("probes, predicts")
    del gp
    _free_device_memory()
    lap("free")
    # (round 4) the same model through the BLOCKED fp32 handle — the per-rank code of the 8-GPU run — on one rank
    from gpmi355x import dist as gd

    gb = gd.ShardedGPE(x, y, g.MeanZero(), g.from_spec(spec), log_noise, dtype=np.float32)
    lap("fp32 blocked fit")
    mub, s2b = gb.predict_f(xs)
    mllb, ab = gb.mll, np.asarray(gb.alpha, dtype=np.float64)
    del gb
    _free_device_memory()
    lap("predict, free")
    # ... and DIRECTLY against fp64 at this size: packed storage holds N = 200 000 fp64 in ~180 GB on the one device (the dense fp64
    # matrix would need 320 GB); the fp64 fit certifies itself by its solve residual on the oracle-rebuilt rows
    g64 = g.GP(x, y, g.MeanZero(), g.from_spec(spec), log_noise, packed=True, stripe_blocks=8)
    lap("fp64 packed fit")
    a64 = np.asarray(g64.alpha, dtype=np.float64)
    r64 = Krows @ a64 + nv * a64[rows] - y[rows]
    assert np.abs(r64).max() <= 1e-8 * np.abs(y).max()
    mu64, s264 = g64.predict_f(xs)
    lap("residual, predict")
    lap.done()
    print(f"[C4 N=200000 d=16] mll fp32 dense {mll32:.3f} / fp32 blocked {mllb:.3f} / fp64 packed {g64.mll:.3f} "
          f"(rel {abs(mll32 / g64.mll - 1):.2e}, {abs(mllb / g64.mll - 1):.2e}); max|dmu| {np.abs(mu32 - mu64).max():.2e} / {np.abs(mub - mu64).max():.2e}, "
          f"max|ds2| {np.abs(s232 - s264).max():.2e} / {np.abs(s2b - s264).max():.2e}, fp64 residual {np.abs(r64).max():.1e}")
    for mll_, a_, mu_, s2_ in ((mll32, a, mu32, s232), (mllb, ab, mub, s2b)):
        assert mll_ == pytest.approx(g64.mll, rel=1e-2)                                       # north_star's fp32 bar
        np.testing.assert_allclose(mu_, mu64, rtol=1e-2, atol=1e-2 * np.abs(mu64).max())
        np.testing.assert_allclose(s2_, s264, rtol=1e-2, atol=1e-2 * np.abs(s264).max())
        np.testing.assert_allclose(a_, a64, rtol=0, atol=1e-2 * np.abs(a64).max())
