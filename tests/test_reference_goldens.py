"""Parity against numbers the REFERENCE produced (tests/golden/reference_transcripts.py): documentation transcripts
`docs/src/Regression.md`, `docs/src/sparse_example.md` and the stored FITC value of `test/test_sparse.jl:156`.

Three layers, each asserted:
  1. the regenerated Julia RNG stream reproduces the inputs the transcripts print (x, y, mean(Y), quantiles) —
     oracle/julia_mt.py is right;
  2. the CPU oracle reproduces the reference's printed results on those inputs (mll, predictive mean / variance,
     optimum) — oracle/gp_oracle.py is pinned by reference outputs;
  3. (-m gpu) the HIP path reproduces the same printed results directly — reference vs device, no oracle in between.
Tolerances are the printed precision (half a unit of the last printed digit) unless the reference states its own
(`atol=1e-3` for the stored FITC value).
"""
import math

import numpy as np
import pytest
from scipy.optimize import minimize

from golden import reference_transcripts as RT
from oracle import gp_oracle as G


def _half_ulp(v):
    """half a unit of the last digit Julia's 6-significant-digit `show` prints for v"""
    if v == 0:
        return 5e-7
    return 0.5 * 10.0 ** (math.floor(math.log10(abs(v))) - 5) * 1.0001


def _assert_printed(got, printed):
    for g_, p in zip(np.asarray(got, dtype=float).ravel(), np.asarray(printed, dtype=float).ravel()):
        assert abs(g_ - p) <= _half_ulp(p), (g_, p)


# ---------------------------------------------------------------------------------------------------------------------
# 1. the stream
# ---------------------------------------------------------------------------------------------------------------------
def test_julia_stream_reproduces_the_regression_inputs():
    x, y = RT.regression_1d()
    for i, v in RT.REG1["printed_x"].items():
        assert abs(x[i] - v) <= 5.1e-6
    _assert_printed(y, RT.REG1["printed_y"])
    assert abs(math.exp(2 * RT.REG1["log_noise"]) - RT.REG1["noise_variance"]) < 1e-15


def test_julia_stream_position_of_the_2d_example_is_found_and_confirmed():
    assert RT.find_regression_2d_offset() == RT.REG2_STREAM_OFFSET
    x, y = RT.regression_2d()
    for (i, j), v in RT.REG2["printed_x"].items():
        assert abs(x[i, j] - v) <= 5.1e-6
    _assert_printed(y[:10], RT.REG2["printed_y_head"])
    _assert_printed(y[-10:], RT.REG2["printed_y_tail"])


def test_julia_beta_and_normal_samplers_reproduce_the_sparse_inputs():
    x, Y = RT.sparse_data(RT.SPARSE["n_doc"])
    for i, v in RT.SPARSE["printed_x"].items():
        assert abs(x[i] - v) <= 5.1e-6
    _assert_printed(Y[:10], RT.SPARSE["printed_y_head"])
    _assert_printed(Y[-10:], RT.SPARSE["printed_y_tail"])  # the LAST draws: every accept/reject branch before them was right
    assert abs(Y.mean() - RT.SPARSE["printed_mean_const"]) <= 5.1e-6
    xu = RT.inducing(x)
    for i, v in RT.SPARSE["printed_inducing"].items():
        assert abs(xu[i] - v) <= 5.1e-6
    # the test-suite's data set is the first 1000 Beta draws of the same stream (then its own noise draws)
    x1, _ = RT.sparse_data(RT.SPARSE["n_test"])
    np.testing.assert_array_equal(x1, x[:1000])


# ---------------------------------------------------------------------------------------------------------------------
# 2. the oracle against the reference's printed results
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_matches_regression_md_1d_mll_and_predict_y():
    x, y = RT.regression_1d()
    fit = G.update_mll(RT.REG1["spec"], x[None, :], y, RT.REG1["log_noise"])
    assert abs(fit["mll"] - RT.REG1["mll"]) <= 5.1e-4
    xs = np.linspace(0.0, 2.0 * np.pi, 100)[None, :]
    mu, s2 = G.predict_y(RT.REG1["spec"], x[None, :], fit, xs, RT.REG1["log_noise"])
    _assert_printed(mu[:10], RT.REG1["predict_y_mu_head"])
    _assert_printed(mu[-10:], RT.REG1["predict_y_mu_tail"])
    _assert_printed(s2[:10], RT.REG1["predict_y_var_head"])
    _assert_printed(s2[-10:], RT.REG1["predict_y_var_tail"])


def test_oracle_optimum_matches_regression_md_1d():
    x, y = RT.regression_1d()

    def f(p):
        spec = ("se_iso", p[1], p[2])
        ft = G.update_mll(spec, x[None, :], y, p[0])
        d = G.update_dmll(spec, x[None, :], y, p[0], fit=ft)
        return -ft["mll"], -np.asarray(d["dmll"])

    res = minimize(f, [-1.0, 0.0, 0.0], jac=True, method="BFGS", options=dict(gtol=1e-9))
    assert abs(res.fun - RT.REG1["opt_minimum"]) <= 5.1e-7
    np.testing.assert_allclose(res.x[:2], RT.REG1["opt_minimizer_head"], rtol=0, atol=2e-6)
    # the printed minimizer is a stationary point of the oracle's mll (|g| = 5.69e-08 in the transcript)
    g0 = f(list(RT.REG1["opt_minimizer_head"]) + [res.x[2]])[1]
    assert np.abs(g0).max() < 1e-5


def test_oracle_matches_regression_md_2d_composite_mll_and_optimum():
    x, y = RT.regression_2d()
    fit = G.update_mll(RT.REG2["spec"], x, y, RT.REG2["log_noise"])
    assert abs(fit["mll"] - RT.REG2["mll"]) <= 5.1e-4

    def mk(p):
        return ("sum", ("mat52_ard", [p[1], p[2]], p[3]), ("se_iso", p[4], p[5]))

    def f(p):
        try:
            ft = G.update_mll(mk(p), x, y, p[0])
            d = G.update_dmll(mk(p), x, y, p[0], fit=ft)
        except G.NotPosDef:
            return 1e10, np.zeros(6)
        return -ft["mll"], -np.asarray(d["dmll"])

    res = minimize(f, [-2.0, 0.0, 0.0, 0.0, 0.0, 0.0], jac=True, method="L-BFGS-B")
    assert abs(res.fun - RT.REG2["opt_minimum"]) <= 5.1e-5
    assert abs(res.x[0] - RT.REG2["opt_minimizer_head"][0]) <= 1e-3  # flat ridge in the vanished Matern term's parameters


def test_oracle_matches_sparse_example_exact_mll_n5000():
    x, Y = RT.sparse_data(RT.SPARSE["n_doc"])
    fit = G.update_mll(RT.SPARSE["spec"], x[None, :], Y, RT.SPARSE["log_noise"], mspec=("const", Y.mean()))
    assert abs(fit["mll"] - RT.SPARSE["exact_mll_n5000"]) <= 5.1e-4  # 8 significant digits


def test_oracle_matches_test_sparse_jl_fitc_value():
    x, Y = RT.sparse_data(RT.SPARSE["n_test"])
    xu = RT.inducing(x)
    ms = ("const", Y.mean())
    f64 = G.fitc_update_mll(RT.SPARSE["spec"], x[None, :], xu[None, :], Y, RT.SPARSE["log_noise"], mspec=ms)
    ext = G.fitc_update_mll_extended(RT.SPARSE["spec"], x[None, :], xu[None, :], Y, RT.SPARSE["log_noise"], mspec=ms)
    assert abs(f64["mll"] - RT.SPARSE["fitc_mll_n1000"]) <= RT.SPARSE["fitc_atol"]   # the reference's own tolerance
    assert abs(ext["mll"] - RT.SPARSE["fitc_mll_n1000"]) <= RT.SPARSE["fitc_atol"]
    full = G.update_mll(RT.SPARSE["spec"], x[None, :], Y, RT.SPARSE["log_noise"], mspec=ms)
    assert abs(f64["mll"] - full["mll"]) <= RT.SPARSE["full_vs_sparse_atol"]         # test_sparse.jl:115


# ---------------------------------------------------------------------------------------------------------------------
# 3. the HIP path against the reference's printed results (no oracle in between)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_matches_regression_md_1d():
    import gpmi355x as g

    x, y = RT.regression_1d()
    gp = g.GP(x, y, g.MeanZero(), g.SE(0.0, 0.0), -1.0)
    assert abs(gp.mll - RT.REG1["mll"]) <= 5.1e-4
    mu, s2 = g.predict_y(gp, np.linspace(0.0, 2.0 * np.pi, 100))
    _assert_printed(mu[:10], RT.REG1["predict_y_mu_head"])
    _assert_printed(mu[-10:], RT.REG1["predict_y_mu_tail"])
    _assert_printed(s2[:10], RT.REG1["predict_y_var_head"])
    _assert_printed(s2[-10:], RT.REG1["predict_y_var_tail"])
    # optimize!(gp) from the transcript's starting point reaches the transcript's optimum (device target and gradient)
    res = g.optimize(gp, method="BFGS", options=dict(gtol=1e-8, maxiter=200))
    assert abs(res.fun - RT.REG1["opt_minimum"]) <= 5.1e-7
    np.testing.assert_allclose(gp.get_params()[:2], RT.REG1["opt_minimizer_head"], rtol=0, atol=5e-6)


@pytest.mark.gpu
def test_device_matches_regression_md_2d_composite():
    import gpmi355x as g

    x, y = RT.regression_2d()
    gp = g.GP(x, y, g.MeanZero(), g.Matern(5 / 2, [0.0, 0.0], 0.0) + g.SE(0.0, 0.0), -2.0)
    assert abs(gp.mll - RT.REG2["mll"]) <= 5.1e-4
    res = g.optimize(gp, options=dict(maxiter=500))
    assert abs(res.fun - RT.REG2["opt_minimum"]) <= 5.1e-5


@pytest.mark.gpu
def test_device_matches_sparse_example_exact_mll_n5000():
    import gpmi355x as g

    x, Y = RT.sparse_data(RT.SPARSE["n_doc"])
    gp = g.GPE(x[None, :], Y, g.MeanConst(Y.mean()), g.SEIso(math.log(0.3), math.log(5.0)), math.log(10.0))
    assert abs(gp.mll - RT.SPARSE["exact_mll_n5000"]) <= 5.1e-4


@pytest.mark.gpu
def test_device_matches_test_sparse_jl_fitc_value():
    import gpmi355x as g

    x, Y = RT.sparse_data(RT.SPARSE["n_test"])
    xu = RT.inducing(x)
    k = g.SEIso(math.log(0.3), math.log(5.0))
    sp = g.FITC(x[None, :], xu[None, :], Y, g.MeanConst(Y.mean()), k, math.log(10.0))
    assert abs(sp.mll - RT.SPARSE["fitc_mll_n1000"]) <= RT.SPARSE["fitc_atol"]
    full = g.GPE(x[None, :], Y, g.MeanConst(Y.mean()), k, math.log(10.0))
    assert abs(sp.mll - full.mll) <= RT.SPARSE["full_vs_sparse_atol"]


# ---------------------------------------------------------------------------------------------------------------------
# 4. the ARD code path (SEArd: the kernel of the bench, C2, C4 and C5) tied to the same printed digits.
#    The reference's SEArd with one length scale per dimension all equal is, entry for entry, its SEIso
#    (src/kernels/se_ard.jl:43 `σ2 exp(-r/2)` on the weighted squared distance against se_iso.jl:39
#    `σ2 exp(-0.5 r/ℓ2)`), so on the transcripts' inputs `SEArd([0.0], 0.0)` must print what `SE(0.0, 0.0)` printed — but the oracle's
#    and the device's ARD leaves are different code from their iso leaves (weighted distance, per-dimension scales,
#    cov_leaf_kernel's pre-scaled inputs), and that code is what these cases pin.
# ---------------------------------------------------------------------------------------------------------------------
SEARD_1D = ("se_ard", [0.0], 0.0)
SEARD_2D_SUM = ("sum", ("mat52_ard", [0.0, 0.0], 0.0), ("se_ard", [0.0, 0.0], 0.0))
SEARD_SPARSE = ("se_ard", [math.log(0.3)], math.log(5.0))


def test_oracle_seard_matches_regression_md_1d_mll_and_predict_y():
    x, y = RT.regression_1d()
    fit = G.update_mll(SEARD_1D, x[None, :], y, RT.REG1["log_noise"])
    assert abs(fit["mll"] - RT.REG1["mll"]) <= 5.1e-4  # docs/src/Regression.md:63
    xs = np.linspace(0.0, 2.0 * np.pi, 100)[None, :]
    mu, s2 = G.predict_y(SEARD_1D, x[None, :], fit, xs, RT.REG1["log_noise"])  # docs/src/Regression.md:83-89
    _assert_printed(mu[:10], RT.REG1["predict_y_mu_head"])
    _assert_printed(mu[-10:], RT.REG1["predict_y_mu_tail"])
    _assert_printed(s2[:10], RT.REG1["predict_y_var_head"])
    _assert_printed(s2[-10:], RT.REG1["predict_y_var_tail"])


def test_oracle_seard_matches_regression_md_2d_composite_mll_and_optimum():
    x, y = RT.regression_2d()
    fit = G.update_mll(SEARD_2D_SUM, x, y, RT.REG2["log_noise"])
    assert abs(fit["mll"] - RT.REG2["mll"]) <= 5.1e-4  # docs/src/Regression.md:332

    # the optimum of :347-351 is reached with the two ARD length scales tied (the iso kernel's one parameter)
    def mk(p):
        return ("sum", ("mat52_ard", [p[1], p[2]], p[3]), ("se_ard", [p[4], p[4]], p[5]))

    def f(p):
        try:
            ft = G.update_mll(mk(p), x, y, p[0])
            d = np.asarray(G.update_dmll(mk(p), x, y, p[0], fit=ft)["dmll"])  # noise, mat52 (2 + 1), se_ard (2 + 1)
        except G.NotPosDef:
            return 1e10, np.zeros(6)
        return -ft["mll"], -np.array([d[0], d[1], d[2], d[3], d[4] + d[5], d[6]])

    res = minimize(f, [-2.0, 0.0, 0.0, 0.0, 0.0, 0.0], jac=True, method="L-BFGS-B")
    assert abs(res.fun - RT.REG2["opt_minimum"]) <= 5.1e-5


def test_oracle_seard_matches_test_sparse_jl_fitc_value():
    x, Y = RT.sparse_data(RT.SPARSE["n_test"])
    xu = RT.inducing(x)
    ms = ("const", Y.mean())
    f64 = G.fitc_update_mll(SEARD_SPARSE, x[None, :], xu[None, :], Y, RT.SPARSE["log_noise"], mspec=ms)
    assert abs(f64["mll"] - RT.SPARSE["fitc_mll_n1000"]) <= RT.SPARSE["fitc_atol"]  # test/test_sparse.jl:156


@pytest.mark.gpu
def test_device_seard_matches_regression_md_1d():
    import gpmi355x as g

    x, y = RT.regression_1d()
    gp = g.GP(x, y, g.MeanZero(), g.SEArd([0.0], 0.0), -1.0)
    assert abs(gp.mll - RT.REG1["mll"]) <= 5.1e-4
    mu, s2 = g.predict_y(gp, np.linspace(0.0, 2.0 * np.pi, 100))
    _assert_printed(mu[:10], RT.REG1["predict_y_mu_head"])
    _assert_printed(mu[-10:], RT.REG1["predict_y_mu_tail"])
    _assert_printed(s2[:10], RT.REG1["predict_y_var_head"])
    _assert_printed(s2[-10:], RT.REG1["predict_y_var_tail"])


@pytest.mark.gpu
def test_device_seard_matches_regression_md_2d_composite():
    import gpmi355x as g

    x, y = RT.regression_2d()
    gp = g.GP(x, y, g.MeanZero(), g.Matern(5 / 2, [0.0, 0.0], 0.0) + g.SEArd([0.0, 0.0], 0.0), -2.0)
    assert abs(gp.mll - RT.REG2["mll"]) <= 5.1e-4
    # the device gradient at the transcript's start: the two ARD components sum to the iso kernel's length-scale derivative
    gp.update_dmll()
    iso = g.GP(x, y, g.MeanZero(), g.Matern(5 / 2, [0.0, 0.0], 0.0) + g.SE(0.0, 0.0), -2.0)
    iso.update_dmll()
    d, di = np.asarray(gp.dmll), np.asarray(iso.dmll)
    np.testing.assert_allclose([d[0], d[1], d[2], d[3], d[4] + d[5], d[6]], di, rtol=1e-9, atol=1e-9 * np.abs(di).max())


@pytest.mark.gpu
def test_device_seard_matches_test_sparse_jl_fitc_value():
    import gpmi355x as g

    x, Y = RT.sparse_data(RT.SPARSE["n_test"])
    xu = RT.inducing(x)
    k = g.SEArd([math.log(0.3)], math.log(5.0))
    sp = g.FITC(x[None, :], xu[None, :], Y, g.MeanConst(Y.mean()), k, math.log(10.0))
    assert abs(sp.mll - RT.SPARSE["fitc_mll_n1000"]) <= RT.SPARSE["fitc_atol"]
