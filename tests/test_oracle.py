"""Pins the CPU oracle (oracle/) — the checker every GPU parity test relies on.

Numbers the reference produced itself (documentation transcripts, the stored FITC
value) are checked in tests/test_reference_goldens.py.  For everything those do not
exercise the oracle is pinned here by (i) the relational properties the reference's own tests
assert, (ii) scikit-learn as an independent implementation of the same maths,
(iii) closed forms, (iv) agreement of the two restatements (NumPy vs scalar C).
"""
import math

import numpy as np
import pytest
import scipy.linalg as sla

from oracle import c_oracle
from oracle import gp_oracle as G
from kernel_cases import ALL, COMPOSITES, D, LEAVES, ids


def _data(n=6, n2=3, d=D, seed=1):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((d, n)), rng.standard_normal((d, n2))


# --- test/kernels.jl:39-41  cov(k,X)[i,j] ≈ cov(k, x_i, x_j) ------------------
@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_cov_matches_pairwise_definition_symmetric(spec):
    X, _ = _data()
    K = G.cov(spec, X)
    n = X.shape[1]
    for i in range(n):
        for j in range(n):
            assert K[i, j] == pytest.approx(G.cov_scalar(spec, list(X[:, i]), list(X[:, j])), rel=1e-13, abs=1e-300)
    assert np.array_equal(K, K.T)


# --- test/kernels.jl:55-60  cov(k,X,X2)[i,j] ≈ cov(k, x_i, x2_j) -------------
@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_cov_matches_pairwise_definition_rect(spec):
    X, X2 = _data()
    K = G.cov(spec, X, X2)
    assert K.shape == (X.shape[1], X2.shape[1])
    for i in range(X.shape[1]):
        for j in range(X2.shape[1]):
            assert K[i, j] == pytest.approx(G.cov_scalar(spec, list(X[:, i]), list(X2[:, j])), rel=1e-13, abs=1e-300)


# --- the two restatements agree (NumPy vectorised vs scalar C loops) ---------
@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_numpy_and_c_oracle_agree(spec):
    X, X2 = _data(n=40, n2=17)
    np.testing.assert_allclose(c_oracle.cov(spec, X), G.cov(spec, X), rtol=2e-14, atol=1e-300)
    np.testing.assert_allclose(c_oracle.cov(spec, X, X2), G.cov(spec, X, X2), rtol=2e-14, atol=1e-300)


def test_c_oracle_assemble_adds_nugget():
    X, _ = _data(n=20)
    spec = ("se_ard", [0.1, 0.2, 0.3], 0.0)
    A = c_oracle.assemble(spec, X, -1.0)
    np.testing.assert_allclose(A, G.cov(spec, X) + math.exp(-2.0) * np.eye(20), rtol=1e-14)
    ln = np.linspace(-2, -1, 20)
    A = c_oracle.assemble(spec, X, ln)
    np.testing.assert_allclose(A, G.cov(spec, X) + np.diag(np.exp(2 * ln)), rtol=1e-14)


# --- noise.jl:31-37: δ is isapprox on coordinates, not index equality --------
def test_noise_kernel_isapprox_semantics():
    X = np.array([[0.5, 0.5, 0.5 * (1 + 1e-9), 0.7], [1.0, 1.0, 1.0, 1.0]])
    K = G.cov(("noise", 0.0), X)
    expect = np.array([[1, 1, 1, 0], [1, 1, 1, 0], [1, 1, 1, 0], [0, 0, 0, 1]], dtype=float)
    np.testing.assert_array_equal(K, expect)
    np.testing.assert_array_equal(c_oracle.cov(("noise", 0.0), X), expect)
    X2 = np.array([[0.5 * (1 + 1e-7)], [1.0]])  # outside rtol sqrt(eps)
    assert G.cov(("noise", 0.0), X, X2)[0, 0] == 0.0


# --- closed forms --------------------------------------------------------------
def test_closed_form_n1_and_n2():
    # N = 1: mll = -(y²/(s2+n2) + log(s2+n2) + log 2π)/2
    x = np.array([[0.3]])
    y = np.array([0.7])
    s2, n2 = math.exp(2 * 0.2), math.exp(2 * -1.0)
    fit = G.update_mll(("se_iso", 0.0, 0.2), x, y, -1.0)
    v = s2 + n2
    assert fit["mll"] == pytest.approx(-(0.49 / v + math.log(v) + G.LOG2PI) / 2, rel=1e-14)
    # N = 2 SEIso: explicit 2×2 inverse / determinant
    x = np.array([[0.0, 1.0]])
    y = np.array([1.0, -0.5])
    k01 = s2 * math.exp(-0.5 * 1.0 / 1.0)
    a, b = s2 + n2, k01
    det = a * a - b * b
    quad = (a * y[0] ** 2 - 2 * b * y[0] * y[1] + a * y[1] ** 2) / det
    fit = G.update_mll(("se_iso", 0.0, 0.2), x, y, -1.0)
    assert fit["mll"] == pytest.approx(-(quad + math.log(det) + 2 * G.LOG2PI) / 2, rel=1e-13)
    np.testing.assert_allclose(fit["alpha"], np.array([a * y[0] - b * y[1], -b * y[0] + a * y[1]]) / det, rtol=1e-13)


# --- independent implementation: scikit-learn --------------------------------
def _sk_kernel(spec):
    from sklearn.gaussian_process import kernels as SK

    name = spec[0]
    if name == "sum":
        return _sk_kernel(spec[1]) + _sk_kernel(spec[2])
    if name == "prod":
        return _sk_kernel(spec[1]) * _sk_kernel(spec[2])
    if name == "noise":
        return SK.WhiteKernel(noise_level=math.exp(2 * spec[1]))
    if name == "const":
        return SK.ConstantKernel(math.exp(2 * spec[1]))
    amp = SK.ConstantKernel(math.exp(2 * spec[2]))
    ls = math.exp(spec[1]) if name.endswith("_iso") else np.exp(np.asarray(spec[1]))
    if name.startswith("se"):
        return amp * SK.RBF(length_scale=ls)
    if name.startswith("mat12"):
        return amp * SK.Matern(length_scale=ls, nu=0.5)
    if name.startswith("mat32"):
        return amp * SK.Matern(length_scale=ls, nu=1.5)
    if name.startswith("mat52"):
        return amp * SK.Matern(length_scale=ls, nu=2.5)
    if name == "rq_iso":
        return amp * SK.RationalQuadratic(length_scale=ls, alpha=math.exp(spec[3]))
    raise KeyError(name)


SK_CASES = [s for s in LEAVES if s[0] not in ("rq_ard", "noise")] + [
    ("sum", ("se_iso", 0.3, 0.1), ("rq_iso", 0.3, 0.2, 0.4)),
    ("prod", ("se_iso", 0.3, 0.1), ("mat12_iso", 0.2, -0.1)),
    ("sum", ("se_ard", [0.3, -0.2, 0.5], 0.0), ("mat52_iso", math.log(0.7), math.log(0.5))),
]


@pytest.mark.parametrize("spec", SK_CASES, ids=ids(SK_CASES))
def test_vs_sklearn_mll_and_predict(spec):
    from sklearn.gaussian_process import GaussianProcessRegressor

    rng = np.random.default_rng(7)
    n, p = 40, 9
    x = rng.uniform(size=(D, n))
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(size=(D, p))
    log_noise = math.log(0.3)
    gpr = GaussianProcessRegressor(kernel=_sk_kernel(spec), alpha=math.exp(2 * log_noise), optimizer=None).fit(x.T, y)
    fit = G.update_mll(spec, x, y, log_noise)
    assert fit["mll"] == pytest.approx(gpr.log_marginal_likelihood_value_, rel=1e-12)
    np.testing.assert_allclose(fit["alpha"], gpr.alpha_, rtol=1e-9)
    mu, s2 = G.predict_f(spec, x, fit, xs)
    mu_sk, cov_sk = gpr.predict(xs.T, return_cov=True)
    np.testing.assert_allclose(mu, mu_sk, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(s2, np.diag(cov_sk), rtol=1e-8, atol=1e-12)
    _, Sig = G.predict_f(spec, x, fit, xs, full_cov=True)
    np.testing.assert_allclose(Sig, cov_sk, rtol=1e-8, atol=1e-11)


# --- test/gp.jl:47-53: predict at the training inputs ≈ y; σ² ≡ diag(full cov);
#     and the per-point loop of GP.jl:69-77 ≡ the batched computation ----------
@pytest.mark.parametrize("spec", [LEAVES[1], LEAVES[7], COMPOSITES[2]], ids=["se_ard", "mat52_ard", "sum+noise"])
def test_predict_properties(spec):
    rng = np.random.default_rng(3)
    n = 10
    x = 2 * np.pi * rng.uniform(size=(D, n))
    y = np.sin(x.sum(axis=0)) + 0.05 * rng.standard_normal(n)
    fit = G.update_mll(spec, x, y, -2.0)
    mu, s2 = G.predict_y(spec, x, fit, x, -2.0)
    np.testing.assert_allclose(mu, y, atol=0.1)  # gp.jl:50
    mu_f, s2_f = G.predict_f(spec, x, fit, x)
    _, Sig = G.predict_f(spec, x, fit, x, full_cov=True)
    np.testing.assert_allclose(s2_f, np.maximum(np.diag(Sig), 0.0), rtol=1e-9, atol=1e-13)  # gp.jl:52
    mu_p, s2_p = G.predict_f(spec, x, fit, x, pointwise=True)
    np.testing.assert_allclose(mu_p, mu_f, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(s2_p, s2_f, rtol=1e-9, atol=1e-13)


def test_mean_functions_and_arg_errors():
    X, _ = _data()
    np.testing.assert_array_equal(G.mean(("zero",), X), np.zeros(6))
    np.testing.assert_array_equal(G.mean(("const", 1.5), X), np.full(6, 1.5))
    np.testing.assert_allclose(G.mean(("lin", [1.0, 2.0, 3.0]), X), X.T @ np.array([1.0, 2.0, 3.0]))
    with pytest.raises(ValueError):
        G.update_mll(("se_iso", 0.0, 0.0), X, np.zeros(5), -1.0)  # GPE.jl:42
    with pytest.raises(ValueError):
        G.cov(("se_iso", 0.0, 0.0), X, np.zeros((2, 4)))  # kernels.jl:34


def test_not_posdef_reports_pivot():
    # duplicate points, Const kernel, no noise ⇒ rank-1 matrix: dpotrf fails at pivot 2
    x = np.zeros((1, 3))
    with pytest.raises(G.NotPosDef) as ei:
        G.update_mll(("const", 0.0), x, np.ones(3), -400.0)
    assert ei.value.info == 2


def test_heteroscedastic_noise_path():
    rng = np.random.default_rng(5)
    x = rng.uniform(size=(2, 30))
    y = rng.standard_normal(30)
    ln = rng.uniform(-2, -1, size=30)
    fit = G.update_mll(("se_iso", -0.5, 0.0), x, y, ln)
    K = G.cov(("se_iso", -0.5, 0.0), x) + np.diag(np.exp(2 * ln))
    np.testing.assert_allclose(K @ fit["alpha"], y, rtol=1e-9, atol=1e-11)
    assert np.isfinite(fit["mll"])  # heteroscedastic.jl:34-48


# --- gradient path: test/kernels.jl:84-93,148-164 — analytic dtarget vs finite differences (rtol 1e-3 there) ----
def _perturb(spec, p, h):
    """spec with its p-th log-parameter (get_params order) shifted by h; returns (new_spec, consumed)."""
    name = spec[0]
    if name in ("sum", "prod"):
        n1 = G.num_params(spec[1])
        if p < n1:
            return (name, _perturb(spec[1], p, h), spec[2])
        return (name, spec[1], _perturb(spec[2], p - n1, h))
    if name == "masked":
        return (name, _perturb(spec[1], p, h), spec[2])
    if name == "fixed":
        return (name, _perturb(spec[1], spec[2][p], h), spec[2])
    if name in ("noise", "const"):
        return (name, spec[1] + h)
    if name.endswith("_iso"):
        vals = list(spec[1:])
        vals[p] += h
        return (name, *vals)
    lls = list(spec[1])
    rest = list(spec[2:])
    if p < len(lls):
        lls[p] += h
    else:
        rest[p - len(lls)] += h
    return (name, lls, *rest)


@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_oracle_gradient_vs_finite_differences(spec):
    rng = np.random.default_rng(9)
    n = 25
    x = rng.uniform(size=(D, n))
    x[:, 3] = x[:, 11]  # a coincident pair: exercises the r = 0 branches of the Matern derivatives
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    ln = math.log(0.3)
    g = G.update_dmll(spec, x, y, ln, ("const", 0.2))
    h = 1e-6
    fd_noise = (G.update_mll(spec, x, y, ln + h, ("const", 0.2))["mll"] - G.update_mll(spec, x, y, ln - h, ("const", 0.2))["mll"]) / (2 * h)
    assert g["dnoise"] == pytest.approx(fd_noise, rel=1e-5, abs=1e-6)
    fd_mean = (G.update_mll(spec, x, y, ln, ("const", 0.2 + h))["mll"] - G.update_mll(spec, x, y, ln, ("const", 0.2 - h))["mll"]) / (2 * h)
    assert g["dmean"][0] == pytest.approx(fd_mean, rel=1e-5, abs=1e-6)
    for p in range(G.num_params(spec)):
        up = G.update_mll(_perturb(spec, p, h), x, y, ln, ("const", 0.2))["mll"]
        dn = G.update_mll(_perturb(spec, p, -h), x, y, ln, ("const", 0.2))["mll"]
        assert g["dkern"][p] == pytest.approx((up - dn) / (2 * h), rel=2e-5, abs=2e-6), f"param {p}"


# --------------------------------------------------------------------------------------------
# FITC oracle: the relational checks test/test_sparse.jl makes on a sparse PDMat
# --------------------------------------------------------------------------------------------
def _fitc_case(n=300, d=2, m=25, seed=5):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(d, n))
    xu = rng.uniform(size=(d, m))
    y = np.sin(4.0 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    return x, xu, y


@pytest.mark.parametrize("spec", [("se_iso", math.log(0.3), 0.2), ("se_ard", [math.log(0.3), math.log(0.5)], 0.0),
                                  ("sum", ("mat52_iso", math.log(0.4), 0.1), ("rq_iso", 0.0, -0.5, 0.3))],
                         ids=["se_iso", "se_ard", "mat52+rq"])
def test_fitc_pdmat_semantics_match_the_dense_matrix(spec):
    """test_sparse.jl:117-132: logdet / `\\` of the sparse PDMat against Matrix(cK)."""
    x, xu, y = _fitc_case()
    f = G.fitc_update_mll(spec, x, xu, y, math.log(0.2), ("const", 0.3))
    S = G.fitc_dense(f)
    c = sla.cho_factor(S, lower=False)
    ym = y - 0.3
    # (the reference's two 1e-10 nuggets sit inside f but not in Matrix(cK): agreement to ~1e-5 of the scale, as
    #  test_sparse.jl's own atol 1e-3 allows for)
    ref_alpha = sla.cho_solve(c, ym)
    np.testing.assert_allclose(f["alpha"], ref_alpha, rtol=1e-4, atol=1e-5 * np.abs(ref_alpha).max())
    logdet = 2.0 * np.sum(np.log(np.diag(c[0])))
    assert abs(f["logdet"] - logdet) < 1e-5 * max(1.0, abs(logdet))
    mll = -(ym @ sla.cho_solve(c, ym) + logdet + x.shape[1] * math.log(2 * math.pi)) / 2
    assert abs(f["mll"] - mll) < 1e-5 * max(1.0, abs(mll))
    # Λ is what makes diag(Σ) exact: diag(Kfu Kuu^-1 Kuf + Λ) = k(x, x) + σ²
    np.testing.assert_allclose(np.diag(S), G._kdiag(spec, x) + math.exp(2 * math.log(0.2)), rtol=1e-9)


def test_fitc_predictions_match_the_dense_conditional():
    """test_sparse.jl:38-111: μ = Qxf (Qff + Λ)^-1 (y - m) + m(x*),  Σ = Kxx - Qxf (Qff + Λ)^-1 Qfx."""
    spec = ("se_ard", [math.log(0.3), math.log(0.4)], 0.1)
    x, xu, y = _fitc_case(n=400, m=30, seed=7)
    f = G.fitc_update_mll(spec, x, xu, y, math.log(0.15))
    xs = np.random.default_rng(8).uniform(size=(2, 9))
    mu, S = G.fitc_predict_f(spec, xu, f, xs, full_cov=True)
    _, v = G.fitc_predict_f(spec, xu, f, xs)
    Kux = G.cov(spec, xu, xs)
    Qxf = Kux.T @ sla.cho_solve((f["Uuu"], False), f["Kuf"])
    c = sla.cho_factor(G.fitc_dense(f))
    np.testing.assert_allclose(mu, Qxf @ sla.cho_solve(c, y), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(S, G.cov(spec, xs) - Qxf @ sla.cho_solve(c, Qxf.T), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(v, np.maximum(np.diag(S), 0.0), rtol=0, atol=0)


def test_fitc_with_inducing_points_at_the_data_is_the_exact_gp():
    """m = n, inducing = x: Qff = Kff, Λ = σ² I, so FITC reproduces the exact mll (up to the two 1e-10 nuggets)."""
    spec = ("mat32_iso", math.log(0.5), 0.0)
    x, _, y = _fitc_case(n=120, m=5, seed=9)
    f = G.fitc_update_mll(spec, x, x, y, math.log(0.3))
    e = G.update_mll(spec, x, y, math.log(0.3))
    assert abs(f["mll"] - e["mll"]) < 1e-6 * abs(e["mll"])
    np.testing.assert_allclose(f["alpha"], e["alpha"], rtol=1e-5, atol=1e-6)


def test_fitc_extended_precision_oracle_agrees_with_fp64_when_well_conditioned():
    spec = ("mat32_iso", math.log(0.4), 0.1)
    x, xu, y = _fitc_case(n=250, m=15, seed=11)
    a = G.fitc_update_mll(spec, x, xu, y, math.log(0.2), ("const", 0.1))
    b = G.fitc_update_mll_extended(spec, x, xu, y, math.log(0.2), ("const", 0.1))
    assert abs(a["mll"] - b["mll"]) < 1e-9 * abs(b["mll"])
    np.testing.assert_allclose(a["alpha"], b["alpha"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(a["alpha_u"], b["alpha_u"], rtol=1e-6, atol=1e-9)


def test_predict_loo_is_refitting_without_the_point():
    """test_crossvalidation.jl:22-37: the analytic LOO equals predict_y of the model fitted on the other points."""
    rng = np.random.default_rng(13)
    n = 40
    x = rng.uniform(size=(2, n))
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)
    spec = ("mat52_iso", math.log(0.5), 0.0)
    ln = math.log(0.2)
    fit = G.update_mll(spec, x, y, ln)
    mu, s2 = G.predict_loo(fit, y)
    for i in (0, 7, 39):
        keep = np.arange(n) != i
        f_i = G.update_mll(spec, x[:, keep], y[keep], ln)
        m_i, v_i = G.predict_y(spec, x[:, keep], f_i, x[:, i:i + 1], ln)
        assert abs(mu[i] - m_i[0]) < 1e-8 and abs(s2[i] - v_i[0]) < 1e-8


def test_fitc_gradient_restatement_against_central_differences():
    """fitc_update_dmll (dmll_noise / dmll_mean! / dmll_kern! of the FITC strategy, fully_indep_train_conditional.jl:200-257
    over subsetofregressors.jl:219-256) differentiates fitc_update_mll: the reference pins the same relation for its
    strategies in test/test_sparse.jl:161-175 (dmll vs finite differences)."""
    rng = np.random.default_rng(0)
    n, m, d = 300, 25, 3
    x = rng.uniform(size=(d, n))
    xu = x[:, ::12][:, :m].copy()
    y = np.sin(3 * x.sum(axis=0)) + 0.1 * rng.standard_normal(n)

    def spec_of(th):
        return ("sum", ("se_ard", list(th[0:3]), th[3]), ("mat52_iso", th[4], th[5]))

    th0 = np.array([-0.5, -0.3, -0.6, 0.2, -0.4, -0.5])
    ln0, c0 = math.log(0.3), 0.2
    g = G.fitc_update_dmll(spec_of(th0), x, xu, y, ln0, ("const", c0))["dmll"]

    def f(ln, c, th):
        return G.fitc_update_mll(spec_of(th), x, xu, y, ln, ("const", c))["mll"]

    h = 1e-5
    fd = [(f(ln0 + h, c0, th0) - f(ln0 - h, c0, th0)) / (2 * h), (f(ln0, c0 + h, th0) - f(ln0, c0 - h, th0)) / (2 * h)]
    for p in range(6):
        e = np.zeros(6)
        e[p] = h
        fd.append((f(ln0, c0, th0 + e) - f(ln0, c0, th0 - e)) / (2 * h))
    np.testing.assert_allclose(g, np.array(fd), rtol=1e-4, atol=1e-4)
