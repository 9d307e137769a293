"""GPU parity tests: the HIP path (through the C ABI, via the gpmi355x host mirror) against the
CPU oracle on identical seeded inputs.  Tolerances follow BASELINE.json's north_star:
rtol 1e-5 (fp64) / 1e-2 (fp32) on log-mll, posterior mean and variance — the fp64 checks
below are much tighter than that where the arithmetic allows it.
"""
import math
import os

import numpy as np
import pytest
import scipy.linalg as sla

import gpmi355x as g
from kernel_cases import ALL, COMPOSITES, D, LEAVES, ids
from oracle import gp_oracle as G

pytestmark = pytest.mark.gpu


def _data(n, n2=0, d=D, seed=1, scale=1.0):
    rng = np.random.default_rng(seed)
    X = scale * rng.standard_normal((d, n))
    X2 = scale * rng.standard_normal((d, n2)) if n2 else None
    return X, X2


def _close(a, b, rtol, atol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not np.all(err <= tol):
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: max violation at {i}: got {a[i]!r} want {b[i]!r} |err|={err[i]:.3e} "
                             f"(max abs err {err.max():.3e}, n_bad={int((err > tol).sum())}/{err.size})")


# --------------------------------------------------------------------------------------------
# cov!  (test/kernels.jl:39-41,55-60 is the definition of correctness)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_cov_symmetric_fp64(spec):
    X, _ = _data(150)
    K = g.cov(g.from_spec(spec), X)
    _close(K, G.cov(spec, X), 1e-12, 1e-14, "cov(k,X)")
    assert np.array_equal(K, K.T)


@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_cov_rect_fp64(spec):
    X, X2 = _data(131, 77)
    _close(g.cov(g.from_spec(spec), X, X2), G.cov(spec, X, X2), 1e-12, 1e-14, "cov(k,X,X2)")


@pytest.mark.parametrize("spec", [LEAVES[0], LEAVES[1], LEAVES[7], LEAVES[8], COMPOSITES[2]], ids=["se_iso", "se_ard", "mat52_ard", "rq_iso", "sum+noise"])
def test_cov_fp32(spec):
    X, X2 = _data(300, 90)
    X32, X232 = X.astype(np.float32), X2.astype(np.float32)
    K = g.cov(g.from_spec(spec), X32, X232, dtype="float32")
    assert K.dtype == np.float32
    _close(K, G.cov(spec, X32.astype(np.float64), X232.astype(np.float64)), 2e-4, 2e-6, "cov fp32")


@pytest.mark.parametrize("spec", [LEAVES[1], LEAVES[4], LEAVES[7], LEAVES[9], COMPOSITES[2]], ids=["se_ard", "mat32_iso", "mat52_ard", "rq_ard", "sum+noise"])
def test_cov_offset_inputs(spec):
    """ inputs with a large common offset (years, timestamps): the reference differences the RAW inputs (distance.jl:41-106), so
    its rounding error is relative to |x - y|; the single-leaf interior-tile kernel works on pre-scaled copies and must not make it relative to
    |x| — both blocks are centred on a common data point before scaling.  n large enough for interior tiles (64 x 128 / 64 x 256)."""
    X, X2 = _data(700, 450)
    X6, X26 = X + 1e6, X2 + 1e6
    k = g.from_spec(spec)
    _close(g.cov(k, X6, X26), G.cov(spec, X6, X26), 1e-11, 1e-13, "cov(k, X + 1e6, X2 + 1e6)")
    _close(g.cov(k, X6), G.cov(spec, X6), 1e-11, 1e-13, "cov(k, X + 1e6)")
    if "noise" in str(spec):
        return  # Noise's isapprox (rtol sqrt(eps) of the ELEMENT type, noise.jl:31-37) calls points 0.3 apart equal at x ~ 1e3 in Float32
    X3 = (X + 1e3).astype(np.float32)
    X23 = (X2 + 1e3).astype(np.float32)
    K32 = g.cov(k, X3, X23, dtype="float32")
    _close(K32, G.cov(spec, X3.astype(np.float64), X23.astype(np.float64)), 2e-4, 2e-6, "cov fp32 (X + 1e3)")


@pytest.mark.parametrize("d", [1, 2, 5, 8, 13, 16, 20, 40])
def test_cov_dimension_sweep(d):
    """d <= 16 takes the register-resident path, larger d the LDS-streaming path."""
    X, X2 = _data(200, 70, d=d, seed=d)
    spec = ("sum", ("se_ard", list(np.linspace(-0.3, 0.4, d)), 0.1), ("mat32_iso", 0.5, -0.2))
    _close(g.cov(g.from_spec(spec), X, X2), G.cov(spec, X, X2), 1e-12, 1e-14, f"cov d={d}")
    _close(g.cov(g.from_spec(spec), X), G.cov(spec, X), 1e-12, 1e-14, f"cov sym d={d}")


def test_cov_noise_kernel_isapprox_and_duplicates():
    X = np.array([[0.5, 0.5, 0.5 * (1 + 1e-9), 0.7, 0.5 * (1 + 1e-7)], [1.0, 1.0, 1.0, 1.0, 1.0]])
    K = g.cov(g.Noise(0.0), X)
    np.testing.assert_array_equal(K, G.cov(("noise", 0.0), X))
    assert K[0, 2] == 1.0 and K[0, 4] == 0.0


def test_cov_argument_errors():
    X, X2 = _data(5, 4)
    with pytest.raises(g.ArgumentError):
        g.cov(g.SEIso(0.0, 0.0), X, X2[:2])  # kernels.jl:34
    with pytest.raises(g.ArgumentError):
        g.cov(g.SEArd([0.0, 0.0], 0.0), X)  # 2 length scales, d = 3


# --------------------------------------------------------------------------------------------
# fit: update_mll!  — mll, alpha, factor
# --------------------------------------------------------------------------------------------
def _fit_both(spec, x, y, ln, mspec=("zero",), mean=None, dtype=np.float64):
    gp = g.GP(x, y, mean or g.MeanZero(), g.from_spec(spec), ln, dtype=dtype)
    ref = G.update_mll(spec, np.asarray(x, dtype=np.float64), y, ln, mspec)
    return gp, ref


@pytest.mark.parametrize("n", [1, 2, 10, 63, 64, 65, 200, 256, 257, 500, 1000, 1537])
def test_fit_sizes_fp64(n):
    """Every padding / panel-boundary case: n below, at and across the 64 / 256 block edges."""
    rng = np.random.default_rng(n)
    x = rng.uniform(size=(2, n))
    y = np.sin(4 * x[0]) + x[1] + 0.1 * rng.standard_normal(n)
    spec = ("se_ard", [math.log(0.3), math.log(0.5)], 0.0)
    gp, ref = _fit_both(spec, x, y, math.log(0.1))
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-10, abs=1e-9)
    _close(gp.alpha, ref["alpha"], 1e-7, 1e-7 * np.abs(ref["alpha"]).max(), f"alpha n={n}")
    assert gp.cK.logdet() == pytest.approx(ref["logdet"], rel=1e-11, abs=1e-9)


@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_fit_all_kernels_fp64(spec):
    rng = np.random.default_rng(11)
    n = 300
    x = rng.uniform(size=(D, n))
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    gp, ref = _fit_both(spec, x, y, math.log(0.2))
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-10)
    _close(gp.alpha, ref["alpha"], 1e-7, 1e-8 * np.abs(ref["alpha"]).max(), "alpha")


def test_factor_matches_lapack_and_reconstructs():
    rng = np.random.default_rng(5)
    n = 700
    x = rng.uniform(size=(4, n))
    y = rng.standard_normal(n)
    spec = ("sum", ("se_ard", [-0.5, -0.3, -0.6, -0.2], 0.2), ("mat52_iso", -0.4, -0.5))
    gp, ref = _fit_both(spec, x, y, math.log(0.1))
    U = gp.cK.cholfactors()
    assert np.array_equal(np.tril(U, -1), np.zeros_like(U))
    _close(U, ref["U"], 1e-9, 1e-11, "U vs dpotrf")
    _close(U.T @ U, ref["K"], 1e-12, 1e-12, "U'U vs K")


def test_pdmat_surface_solve_whiten_logdet():
    """AbstractPDMat semantics pinned by test/test_sparse.jl:129-132 (tr/logdet/\\ vs dense)."""
    rng = np.random.default_rng(8)
    n = 450
    x = rng.uniform(size=(3, n))
    y = rng.standard_normal(n)
    spec = ("se_iso", -0.7, 0.1)
    gp, ref = _fit_both(spec, x, y, -1.5)
    B = rng.standard_normal((n, 5))
    _close(gp.cK.solve(B), sla.cho_solve((ref["U"], False), B), 1e-7, 1e-8, "cK \\ B")
    _close(gp.cK.solve(B[:, 0]), sla.cho_solve((ref["U"], False), B[:, 0]), 1e-7, 1e-8, "cK \\ b")
    _close(gp.cK.whiten(B), sla.solve_triangular(ref["U"], B, trans="T"), 1e-8, 1e-9, "whiten")
    assert gp.cK.logdet() == pytest.approx(ref["logdet"], rel=1e-12)


def test_means_and_refit_after_param_change():
    """test/gp.jl:67-73 (update after mutating the kernel) + MeanConst / MeanLin."""
    rng = np.random.default_rng(2)
    n = 320
    x = rng.uniform(size=(3, n))
    y = 2.0 + x.T @ np.array([1.0, -2.0, 0.5]) + 0.1 * rng.standard_normal(n)
    spec = ("mat32_ard", [0.1, 0.0, -0.1], 0.3)
    gp, ref = _fit_both(spec, x, y, -1.0, ("lin", [1.0, -2.0, 0.5]), g.MeanLin([1.0, -2.0, 0.5]))
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-10)
    hyp = gp.get_params()
    assert len(hyp) == 1 + 3 + 4
    hyp2 = [h + 0.1 for h in hyp]
    gp.set_params(hyp2)
    gp.update_target()
    spec2 = ("mat32_ard", [0.2, 0.1, 0.0], 0.4)
    ref2 = G.update_mll(spec2, x, y, -0.9, ("lin", [1.1, -1.9, 0.6]))
    assert gp.mll == pytest.approx(ref2["mll"], rel=1e-10)
    assert gp.target == gp.mll


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["fp64", "fp32"])
def test_mean_only_update_replaces_the_device_alpha(dtype):
    """ update_mll!(gp; kern = false, noise = false) after a change of the mean (GPE.jl:203-211 with update_cK! skipped) keeps
    the factor; predict_f and update_dmll! read the DEVICE copy of alpha, which gpmi_update_alpha must replace: compared with a full
    oracle fit at the new mean, then with the device's own full refit."""
    rng = np.random.default_rng(21)
    n = 900
    x = rng.uniform(size=(3, n))
    y = 2.0 + x.T @ np.array([1.0, -2.0, 0.5]) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(3, 40))
    spec = ("sum", ("se_ard", [0.1, 0.0, -0.1], 0.3), ("mat52_iso", -0.2, -0.3))
    gp = g.GP(x, y, g.MeanLin([1.0, -2.0, 0.5]), g.from_spec(spec), -1.0, dtype=dtype)
    b2 = [0.6, -1.5, 0.9]
    gp.mean.set_params(b2)
    gp.update_mll(kern=False, noise=False)
    x64 = np.asarray(gp.x, dtype=np.float64)
    ref = G.update_mll(spec, x64, y, -1.0, ("lin", b2))
    tol = 1e-10 if dtype == np.float64 else 1e-2
    assert gp.mll == pytest.approx(ref["mll"], rel=tol)
    ra, aa = (1e-7, 1e-8) if dtype == np.float64 else (2e-2, 2e-2)
    _close(gp.alpha, ref["alpha"], ra, aa * np.abs(ref["alpha"]).max(), "alpha after the mean-only update")
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(spec, x64, ref, np.asarray(xs, dtype=dtype).astype(np.float64), ("lin", b2))
    rm = 1e-7 if dtype == np.float64 else 1e-2
    _close(mu, mu_o, rm, rm, "predict_f mean after the mean-only update")
    gp.update_dmll()
    d1 = np.array(gp.dmll)
    if dtype == np.float64:
        d_o = G.update_dmll(spec, x64, y, -1.0, ("lin", b2), fit=ref)["dmll"]
        _close(d1, d_o, 1e-6, 1e-8 * np.abs(d_o).max(), "update_dmll after the mean-only update")
    gp.update_mll()  # the full refit must agree with the shortcut
    mu_r, _ = gp.predict_f(xs)
    gp.update_dmll()
    _close(mu, mu_r, rm, rm, "shortcut vs refit: predict_f")
    _close(d1, gp.dmll, 1e-6 if dtype == np.float64 else 5e-2, (1e-8 if dtype == np.float64 else 5e-2) * np.abs(gp.dmll).max(), "shortcut vs refit: dmll")


def test_heteroscedastic_noise():
    """Vector logNoise path of update_cK! (GPE.jl:177-186; test/heteroscedastic.jl:34-48)."""
    rng = np.random.default_rng(4)
    n = 260
    x = rng.uniform(size=(2, n))
    y = rng.standard_normal(n)
    ln = rng.uniform(-2.0, -0.5, size=n)
    spec = ("se_iso", -0.5, 0.0)
    gp, ref = _fit_both(spec, x, y, ln)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-10)
    _close(gp.alpha, ref["alpha"], 1e-7, 1e-9, "alpha")


def test_not_posdef_reports_pivot_and_handle_survives():
    """PosDefException(info) contract (GP.jl:110; optimize.jl:56-58): status + pivot, handle reusable."""
    n = 130
    x = np.zeros((1, n))
    x[0, 70:] = np.linspace(1, 2, n - 70)
    y = np.ones(n)
    gp = g.GP(x, y, g.MeanZero(), g.SEIso(0.0, 0.0), -1.0)  # fine with noise
    gp.set_params([-400.0, 0.0, 0.0])  # no noise, 70 coincident points -> rank deficient
    with pytest.raises(g.PosDefException) as ei:
        gp.update_mll()
    with pytest.raises(G.NotPosDef) as eo:
        G.update_mll(("se_iso", 0.0, 0.0), x, y, -400.0)
    assert ei.value.info == eo.value.info == 2
    with pytest.raises(g.ArgumentError):
        gp.predict_f(x)  # no valid factorisation
    gp.set_params([-1.0, 0.0, 0.0])
    gp.update_mll()  # handle still usable
    ref = G.update_mll(("se_iso", 0.0, 0.0), x, y, -1.0)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-10)


def test_not_posdef_late_pivot():
    """A failure deep inside the third outer panel reports the same 1-based pivot as dpotrf."""
    n = 600
    x = np.arange(n, dtype=np.float64)[None, :]
    x[0, 555] = x[0, 17]  # exact duplicate, everything else far apart (l = 0.05): K ~ I + e_555 e_17'
    y = np.random.default_rng(0).standard_normal(n)
    with pytest.raises(g.PosDefException) as ei:
        g.GP(x, y, g.MeanZero(), g.SEIso(-3.0, 0.0), -400.0)
    with pytest.raises(G.NotPosDef) as eo:
        G.update_mll(("se_iso", -3.0, 0.0), x, y, -400.0)
    assert ei.value.info == eo.value.info == 556


def test_constructor_argument_errors():
    with pytest.raises(g.ArgumentError):
        g.GP(np.zeros((2, 5)), np.zeros(4), g.MeanZero(), g.SEIso(0.0, 0.0), -1.0)  # GPE.jl:42
    gp = g.GP(np.random.default_rng(0).uniform(size=(2, 20)), np.zeros(20), g.MeanZero(), g.SEIso(0.0, 0.0), -1.0)
    with pytest.raises(g.ArgumentError):
        gp.predict_f(np.zeros((3, 4)))  # GP.jl:65


# --------------------------------------------------------------------------------------------
# predict_f / predict_y
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("spec", [LEAVES[1], LEAVES[5], LEAVES[8], COMPOSITES[2], COMPOSITES[3]],
                         ids=["se_ard", "mat32_ard", "rq_iso", "sum+noise", "(se+mat12)*rq"])
@pytest.mark.parametrize("n,p", [(300, 40), (777, 130)])
def test_predict_fp64(spec, n, p):
    rng = np.random.default_rng(n + p)
    x = rng.uniform(size=(D, n))
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(D, p))
    xs[:, :3] = x[:, :3]  # test points equal to training points (Noise kernel fires there)
    gp, ref = _fit_both(spec, x, y, math.log(0.15), ("const", 0.3), g.MeanConst(0.3))
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(spec, x, ref, xs, ("const", 0.3))
    _close(mu, mu_o, 1e-8, 1e-9, "mu")
    _close(s2, s2_o, 1e-7, 1e-10, "sigma2")
    assert np.all(s2 >= 0)
    mu_f, Sig = gp.predict_f(xs, full_cov=True)
    _, Sig_o = G.predict_f(spec, x, ref, xs, ("const", 0.3), full_cov=True)
    _close(mu_f, mu_o, 1e-8, 1e-9, "mu (full_cov)")
    _close(Sig, Sig_o, 1e-7, 1e-10, "Sigma")
    _close(Sig, Sig.T, 0, 1e-13, "Sigma symmetric")
    # test/gp.jl:52: per-point sigma2 == diag(full cov) (up to the clamp)
    _close(s2, np.maximum(np.diag(Sig), 0.0), 1e-8, 1e-11, "sigma2 vs diag(Sigma)")
    mu_y, s2_y = gp.predict_y(xs)
    _close(s2_y, s2_o + math.exp(2 * math.log(0.15)), 1e-7, 1e-10, "predict_y")


def test_predict_at_training_inputs_recovers_y():
    """test/gp.jl:47-50: predict_y(gp, x)[1] ≈ y with atol 0.1 (d = 3, n = 10)."""
    rng = np.random.default_rng(1)
    x = 2 * np.pi * rng.uniform(size=(3, 10))
    y = np.sin(x.sum(axis=0)) + 0.05 * rng.standard_normal(10)
    gp = g.GP(x, y, g.MeanZero(), g.SEArd([0.0, 0.0, 0.0], 0.0), -2.0)
    mu, s2 = gp.predict_y(x)
    np.testing.assert_allclose(mu, y, atol=0.1)


def test_one_dimensional_vector_inputs():
    """GP(x::Vector, y) / predict_f(gp, x::Vector) (GPE.jl:96-97, GP.jl:82); config C1 of BASELINE.json."""
    rng = np.random.default_rng(3)
    n = 500
    x = 2 * np.pi * rng.uniform(size=n)
    y = np.sin(x) + 0.05 * rng.standard_normal(n)
    gp = g.GP(x, y, g.MeanZero(), g.SEIso(0.0, 0.0), -1.0)
    ref = G.update_mll(("se_iso", 0.0, 0.0), x[None, :], y, -1.0)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-10)
    xs = np.linspace(0, 2 * np.pi, 100)
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(("se_iso", 0.0, 0.0), x[None, :], ref, xs[None, :])
    _close(mu, mu_o, 1e-7, 1e-8, "mu")
    _close(s2, s2_o, 1e-6, 1e-9, "sigma2")


# --------------------------------------------------------------------------------------------
# fp32 path (no reference counterpart; north_star tolerance rtol 1e-2 vs the fp64 oracle)
# --------------------------------------------------------------------------------------------
def test_fit_predict_fp32():
    x, y, xs = G.synthetic_inputs(1500, 8, p=64)
    ll = [math.log(0.5) + 0.05 * k for k in range(8)]
    spec = ("se_ard", ll, 0.0)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.1), dtype=np.float32)
    ref = G.update_mll(spec, x, y, math.log(0.1))
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-2)
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(spec, x, ref, xs)
    _close(mu, mu_o, 1e-2, 1e-2, "mu fp32")
    _close(s2, s2_o, 1e-2, 1e-3, "sigma2 fp32")


# --------------------------------------------------------------------------------------------
# golden fixture (reference's simdata.csv inputs, 12 benchmark kernels)
# --------------------------------------------------------------------------------------------
def test_golden_simdata_bench_kernels():
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden

    z = np.load(os.path.join(here, "golden", "simdata_bench_kernels.npz"))
    for i, name in enumerate(z["names"]):
        k = g.from_spec(make_golden.KERNS[str(name)])
        gp = g.GP(z["x"], z["y"], g.MeanConst(float(z["mean_const"])), k, float(z["log_noise"]))
        assert gp.mll == pytest.approx(float(z[f"mll_{i}"]), rel=1e-10), name
        _close(gp.alpha, z[f"alpha_{i}"], 1e-7, 1e-9, f"alpha[{name}]")
        mu, s2 = gp.predict_f(z["xpred"])
        _close(mu, z[f"mu_{i}"], 1e-7, 1e-9, f"mu[{name}]")
        _close(s2, z[f"s2_{i}"], 1e-7, 1e-9, f"s2[{name}]")


# --------------------------------------------------------------------------------------------
# mid-size synthetic workload of SURVEY §8(d) against the oracle (seconds on the host)
# --------------------------------------------------------------------------------------------
def test_synthetic_workload_n4000_vs_oracle():
    x, y, xs = G.synthetic_inputs(4000, 8, p=256)
    ll = [math.log(0.5) + 0.05 * k for k in range(8)]
    spec = ("sum", ("sum", ("se_ard", ll, 0.0), ("mat52_iso", math.log(0.7), math.log(0.5))), ("noise", math.log(0.05)))
    gp, ref = _fit_both(spec, x, y, math.log(0.1))
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-9)
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(spec, x, ref, xs)
    _close(mu, mu_o, 1e-6, 1e-8, "mu")
    _close(s2, s2_o, 1e-5, 1e-9, "sigma2")


# --------------------------------------------------------------------------------------------
# gradient path: update_dmll! on the device vs the oracle's restatement of the reference formulas
# (themselves checked against finite differences, as test/kernels.jl:84-93,148-164 do)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_gradient_all_kernels_fp64(spec):
    rng = np.random.default_rng(21)
    n = 333
    x = rng.uniform(size=(D, n))
    x[:, 7] = x[:, 200]  # coincident pair: r = 0 branches of the Matern derivatives, Noise off-diagonal hit
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    ln = math.log(0.25)
    gp = g.GP(x, y, g.MeanConst(0.2), g.from_spec(spec), ln)
    gp.update_dmll()
    ref = G.update_dmll(spec, x, y, ln, ("const", 0.2))
    scale = np.abs(ref["dmll"]).max() + 1e-12
    _close(gp.dmll, ref["dmll"], 1e-7, 1e-9 * scale, "dmll")
    assert len(gp.dmll) == 1 + 1 + G.num_params(spec)


def test_gradient_switches_and_mean_lin():
    rng = np.random.default_rng(22)
    n = 700  # spans three outer panels
    x = rng.uniform(size=(3, n))
    y = 1.0 + x.T @ np.array([1.0, -2.0, 0.5]) + 0.1 * rng.standard_normal(n)
    spec = ("prod", ("sum", ("se_ard", [0.1, 0.0, -0.1], 0.3), ("mat32_iso", 0.2, -0.2)), ("rq_iso", 0.5, 0.1, 0.3))
    gp = g.GP(x, y, g.MeanLin([1.0, -2.0, 0.5]), g.from_spec(spec), -1.0)
    ref = G.update_dmll(spec, x, y, -1.0, ("lin", [1.0, -2.0, 0.5]))
    gp.update_dmll()
    _close(gp.dmll, ref["dmll"], 1e-7, 1e-9 * np.abs(ref["dmll"]).max(), "dmll")
    gp.update_dmll(noise=False, domean=False)
    _close(gp.dmll, ref["dkern"], 1e-7, 1e-9 * np.abs(ref["dkern"]).max(), "dmll kern only")
    gp.update_mll_and_dmll(kern=True, noise=True, domean=True)
    assert gp.target == gp.mll and len(gp.dtarget) == 1 + 3 + 9


@pytest.mark.parametrize("d,spec_name", [(24, "se_ard"), (32, "sum_se_rq"), (20, "prod_mat_se")])
def test_gradient_beyond_16_inputs_and_48_parameters(d, spec_name):
    """round 3 (VERDICT r2 missing item 5): the register forms of the device gradient cover d <= 32 and <= 64 hyper-parameters
    (dmll_kernel<T, 32>); round 4: beyond either the limit-free form runs (dmll_kernel<T, 0>) — the reference's dmll_kern! has no
    such limit (src/GPE.jl:219-241).  Dense handle and blocked handle, against the oracle (sum_se_rq: 67 parameters)."""
    rng = np.random.default_rng(31)
    n = 420
    x = rng.uniform(size=(d, n))
    y = np.sin(x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    ll = [math.log(1.5) + 0.02 * k for k in range(d)]
    spec = {"se_ard": ("se_ard", ll, 0.1),
            "sum_se_rq": ("sum", ("se_ard", ll, 0.1), ("rq_ard", [v + 0.3 for v in ll], -0.2, 0.4)),          # 33 + 34 = 67 > 64: see below
            "prod_mat_se": ("prod", ("mat52_ard", ll, 0.0), ("se_ard", [v + 0.5 for v in ll], -0.3))}[spec_name]  # 21 + 21 = 42
    ln = math.log(0.2)
    ref = G.update_dmll(spec, x, y, ln)
    for kw in (dict(), dict(packed=True, block=256, stripe_blocks=1)):
        gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), ln, **kw)
        gp.update_dmll()
        _close(gp.dmll, ref["dmll"], 1e-7, 1e-9 * np.abs(ref["dmll"]).max(), "dmll")


@pytest.mark.parametrize("d,spec_name", [(100, "se_ard"), (70, "sum_mat_noise"), (40, "masked_rq")])
def test_no_input_dimension_limit_cov_fit_predict_gradient(d, spec_name):
    """VERDICT r3 missing 3 / next 9: cov! for any d (src/kernels/distance.jl:41-106 loops over whatever `dim` is), and with it fit,
    predict_f and update_dmll! — d = 100 SEArd has 101 kernel parameters.  Beyond d = 64 the covariance interpreter reads its operands
    from global memory instead of LDS (cov.hip COV_GLOBAL_X), beyond d = 32 / 64 hyper-parameters the gradient runs its limit-free
    form (grad.hip); the kernel program's weight tables are sized at run time (common.h DevProgram).  Against the oracle."""
    rng = np.random.default_rng(77)
    n = 500
    x = rng.uniform(size=(d, n))
    y = np.sin(x[:5].sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(d, 40))
    ll = [math.log(2.0) + 0.01 * k for k in range(d)]
    spec = {"se_ard": ("se_ard", ll, 0.1),
            "sum_mat_noise": ("sum", ("sum", ("mat52_ard", ll, 0.0), ("se_iso", math.log(3.0), -0.5)), ("noise", math.log(0.05))),
            "masked_rq": ("sum", ("masked", ("rq_ard", [0.2, 0.3, 0.1, 0.4], -0.2, 0.3), [0, 3, 17, 39]), ("mat32_iso", math.log(4.0), 0.0))}[spec_name]
    ln = math.log(0.2)
    K = g.cov(g.from_spec(spec), x)
    np.testing.assert_allclose(K, G.cov(spec, x), rtol=1e-12, atol=1e-14)
    Kr = g.cov(g.from_spec(spec), x[:, :130], xs)
    np.testing.assert_allclose(Kr, G.cov(spec, x[:, :130], xs), rtol=1e-12, atol=1e-14)
    ref = G.update_mll(spec, x, y, ln)
    dref = G.update_dmll(spec, x, y, ln, fit=ref)
    mu_o, s2_o = G.predict_f(spec, x, ref, xs)
    for kw in (dict(), dict(packed=True, block=256, stripe_blocks=1)):
        gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), ln, **kw)
        assert abs(gp.mll - ref["mll"]) <= 1e-10 * abs(ref["mll"])
        mu, s2 = gp.predict_f(xs)
        _close(mu, mu_o, 1e-7, 1e-9, "mu")
        _close(s2, s2_o, 1e-6, 1e-10, "s2")
        gp.update_dmll()
        _close(gp.dmll, dref["dmll"], 1e-7, 1e-9 * np.abs(dref["dmll"]).max(), "dmll")


def test_gradient_synthetic_d8_n3000():
    x, y, _ = G.synthetic_inputs(3000, 8, p=4)
    ll = [math.log(0.5) + 0.05 * k for k in range(8)]
    spec = ("sum", ("se_ard", ll, 0.0), ("mat52_iso", math.log(0.7), math.log(0.5)))
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), math.log(0.1))
    gp.update_dmll()
    ref = G.update_dmll(spec, x, y, math.log(0.1))
    _close(gp.dmll, ref["dmll"], 1e-6, 1e-8 * np.abs(ref["dmll"]).max(), "dmll")


def test_gradient_n9000_chunked_inverse(monkeypatch):
    """n >= 8192: K^-1 = L^-T L^-1 is accumulated in chunks of 2048 columns of L^-T (api.hip grad_t) — first-touch rows written
    with the sign flipped, the block above them accumulated — and the trace kernel reads -K^-1.  fp64 against the oracle;
    fp32 (whose gradient is a difference of O(1e3) terms at this size: ~5 % on its smallest component either way) against the
    one-product form of the same arithmetic."""
    x, y, _ = G.synthetic_inputs(9000, 4, p=4)
    spec = ("sum", ("se_ard", [math.log(0.4), math.log(0.5), math.log(0.6), math.log(0.7)], 0.0), ("mat32_iso", math.log(0.8), -0.5))
    ln = math.log(0.2)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(spec), ln)
    gp.update_dmll()
    ref = G.update_dmll(spec, x, y, ln)
    _close(gp.dmll, ref["dmll"], 1e-6, 1e-8 * np.abs(ref["dmll"]).max(), "dmll")
    g32 = g.GP(x.astype(np.float32), y, g.MeanZero(), g.from_spec(spec), ln, dtype=np.float32)
    g32.update_dmll()
    monkeypatch.setenv("GPMI_GRAD_CHUNK", "0")
    one = g.GP(x.astype(np.float32), y, g.MeanZero(), g.from_spec(spec), ln, dtype=np.float32, ctx=g.Context(0))
    one.update_dmll()
    scale = np.abs(ref["dmll"]).max()
    _close(g32.dmll, one.dmll, 1e-2, 1e-3 * scale, "fp32 chunked vs one product")
    _close(g32.dmll, ref["dmll"], 1e-1, 1e-2 * scale, "fp32 vs fp64 oracle")


def test_optimize_with_device_gradient_matches_reference_contract():
    """test/optim.jl:20-25 (target improves), :54-82 (switched-off parameter groups stay bit-identical)."""
    x, y, _ = G.synthetic_inputs(400, 2, p=4)
    gp = g.GP(x, y, g.MeanConst(0.0), g.SEArd([0.0, 0.0], 0.0) + g.Noise(-2.0), -1.0)
    t0 = gp.target
    before = gp.get_params()
    res = g.optimize(gp, domean=False, options={"maxiter": 8})
    assert gp.target > t0 + 1.0
    after = gp.get_params()
    assert after[1] == before[1]  # the mean parameter was not optimised
    assert res.nfev < 40           # a gradient-based run, not a finite-difference one


# --------------------------------------------------------------------------------------------
# look-ahead Cholesky / whitening: sizes on both sides of its thresholds, and a pivot that
# fails on the side stream while a trailing update is in flight
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,p", [(1100, 64), (4700, 300), (6100, 1500)])
def test_lookahead_transitions_vs_oracle(n, p):
    x, y, xs = G.synthetic_inputs(n, 4, p=p)
    spec = ("sum", ("se_ard", [math.log(0.4), math.log(0.5), math.log(0.6), math.log(0.7)], 0.0), ("mat32_iso", math.log(0.8), -0.5))
    gp, ref = _fit_both(spec, x, y, math.log(0.15))
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-9)
    _close(gp.alpha, ref["alpha"], 1e-6, 1e-8 * np.abs(ref["alpha"]).max(), "alpha")
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(spec, x, ref, xs)
    _close(mu, mu_o, 1e-6, 1e-8, "mu")
    _close(s2, s2_o, 1e-5, 1e-9, "sigma2")


def test_not_posdef_during_lookahead():
    """The failing pivot (1301) belongs to a diagonal block that is factored on the side stream under a trailing update."""
    n = 6200
    x = np.arange(n, dtype=np.float64)[None, :]
    x[0, 1300] = x[0, 40]
    y = np.random.default_rng(1).standard_normal(n)
    with pytest.raises(g.PosDefException) as ei:
        g.GP(x, y, g.MeanZero(), g.SEIso(-3.0, 0.0), -400.0)
    assert ei.value.info == 1301
    gp = g.GP(x, y, g.MeanZero(), g.SEIso(-3.0, 0.0), -1.0)  # the context is still usable afterwards
    assert np.isfinite(gp.mll)


# --------------------------------------------------------------------------------------------
# predict_LOO / logp_LOO (src/crossvalidation.jl) and rand (src/GP.jl:120-146)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [150, 700, 1500])
def test_predict_loo_matches_the_oracle(n):
    x, y, _ = G.synthetic_inputs(n, 3, p=4)
    spec = ("sum", ("se_ard", [math.log(0.4), math.log(0.5), math.log(0.6)], 0.0), ("mat32_iso", math.log(0.8), -0.5))
    gp, ref = _fit_both(spec, x, y, math.log(0.15))
    mu, s2 = gp.predict_LOO()
    mu_o, s2_o = G.predict_loo(ref, y)
    _close(s2, s2_o, 1e-7, 1e-12, "loo variance")
    _close(mu, mu_o, 1e-7, 1e-9, "loo mean")
    ref_lp = float(np.sum(-0.5 * np.log(2 * np.pi * s2_o) - 0.5 * (y - mu_o) ** 2 / s2_o))
    assert gp.logp_LOO() == pytest.approx(ref_lp, rel=1e-8)


def test_rand_draws_have_the_predictive_moments():
    """test/gp.jl:62-64 draws samples; here their first two moments are checked against predict_f(full_cov=true)."""
    x, y, xs = G.synthetic_inputs(300, 2, p=6)
    gp = g.GP(x, y, g.MeanZero(), g.SEArd([math.log(0.4), math.log(0.5)], 0.0), math.log(0.2))
    mu, S = gp.predict_f(xs, full_cov=True)
    draws = gp.rand(xs, n=40000, rng=np.random.default_rng(3))
    assert draws.shape == (6, 40000)
    np.testing.assert_allclose(draws.mean(axis=1), mu, atol=4 * np.sqrt(np.diag(S).max() / 40000) + 1e-3)
    np.testing.assert_allclose(np.cov(draws), S, atol=0.03 * np.abs(S).max() + 1e-4)


def test_tiny_noise_keeps_lapack_accuracy():
    """logNoise = -8 on a smooth kernel: the factorisation switches to refined block solves and the mll
    agrees with LAPACK's to well inside the 1e-5 parity bound (both are ~1e-9 relative from the 80-bit value)."""
    rng = np.random.default_rng(5)
    n = 1500
    x = rng.uniform(size=(2, n))
    y = np.sin(4 * x.sum(axis=0)) + 0.05 * rng.standard_normal(n)
    spec = ("se_ard", [math.log(0.3), math.log(0.4)], 0.0)
    gp, ref = _fit_both(spec, x, y, -8.0)
    assert abs(gp.mll - ref["mll"]) <= 2e-8 * abs(ref["mll"])


def test_destroyed_model_returns_its_device_memory_after_a_profiled_fit():
    """gpmi_gp_destroy gives the factor's memory back even after fits bracketed by gpmi_profile_enable: the start / stop events
    attached to the trailing-update dispatches pin the kernel command (and with it the factor) for as long as they exist."""
    import gc

    import torch

    ctx = g.Context.default(0)
    rng = np.random.default_rng(3)
    n = 12000  # fp64 factor: 1.15 GB
    x = rng.uniform(size=(4, n))
    y = np.sin(x.sum(axis=0))

    def run(profile):
        gp = g.GP(x, y, g.MeanZero(), g.SEArd([0.0] * 4, 0.0), -1.0, ctx=ctx)
        if profile:
            ctx.profile_enable(True)
        gp.update_mll()
        if profile:
            ctx.profile_get(g._lib.PROF_SYRK)
            ctx.profile_enable(False)

    run(False)  # context scratch reaches its size for this n
    gc.collect()
    free0 = torch.cuda.mem_get_info()[0]
    run(True)
    gc.collect()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) / 1e9:.2f} GB still held after the model was destroyed"
