"""Prior plumbing of update_target! (src/common.jl:118-170, src/GPE.jl:346-392, 514-526; behaviour checked by the
reference in test/optim.jl:37-52): host logic only, so it runs without a device."""
import math
import os
import sys

import numpy as np
import pytest
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianprocesses.jl_amd"))
import gpmi355x as g  # noqa: E402
from gpmi355x import gpe as gpe_mod  # noqa: E402
from gpmi355x import priors as P  # noqa: E402


def test_normal_and_uniform_densities():
    n = g.Normal(0.3, 1.7)
    for x in (-2.0, 0.3, 4.1):
        assert n.logpdf(x) == pytest.approx(stats.norm(0.3, 1.7).logpdf(x), rel=1e-13)
        h = 1e-6
        assert n.gradlogpdf(x) == pytest.approx((n.logpdf(x + h) - n.logpdf(x - h)) / (2 * h), rel=1e-6, abs=1e-9)
    u = g.Uniform(-1.0, 3.0)
    assert u.logpdf(0.0) == pytest.approx(-math.log(4.0)) and u.logpdf(3.5) == -math.inf and u.gradlogpdf(0.0) == 0.0
    sp = P.Scipy(stats.gamma(2.0, scale=0.5))
    assert sp.logpdf(0.8) == pytest.approx(stats.gamma(2.0, scale=0.5).logpdf(0.8))
    assert sp.gradlogpdf(0.8) == pytest.approx(1.0 / 0.8 - 2.0, rel=1e-5)  # d/dx [(k-1) log x - x/theta]
    with pytest.raises(g.ArgumentError):
        g.Normal(0.0, 0.0)


def test_leaf_priors_length_check_and_values():
    k = g.SEArd([0.1, -0.2], 0.3)
    assert g.prior_logpdf(k) == 0.0 and list(g.prior_gradlogpdf(k)) == [0.0, 0.0, 0.0]  # no priors set: common.jl:155,164
    with pytest.raises(g.ArgumentError):
        g.set_priors(k, [g.Normal()])
    pri = [g.Normal(0.0, 1.0), g.Normal(1.0, 2.0), g.Uniform(-1.0, 1.0)]
    g.set_priors(k, pri)
    expect = stats.norm(0, 1).logpdf(0.1) + stats.norm(1, 2).logpdf(-0.2) - math.log(2.0)
    assert g.prior_logpdf(k) == pytest.approx(expect, rel=1e-13)
    np.testing.assert_allclose(g.prior_gradlogpdf(k), [-0.1, (1.0 + 0.2) / 4.0, 0.0], rtol=1e-13)


def test_composites_split_priors_between_components():
    left, right = g.SEIso(0.0, 0.5), g.RQIso(0.1, 0.2, 0.3)
    k = left + g.Masked(right, [0])
    pri = [g.Normal(float(i), 1.0 + i) for i in range(5)]
    g.set_priors(k, pri)  # pair_kernel.jl:30-36, masked_kernel.jl:90
    assert g.get_priors(left) == pri[:2] and g.get_priors(right) == pri[2:] and g.get_priors(k) == pri
    theta = k.get_params()
    assert g.prior_logpdf(k) == pytest.approx(sum(p.logpdf(t) for p, t in zip(pri, theta)), rel=1e-13)
    np.testing.assert_allclose(g.prior_gradlogpdf(k), [p.gradlogpdf(t) for p, t in zip(pri, theta)], rtol=1e-13)
    m = g.MeanConst(0.4) + g.MeanLin([1.0, -1.0])
    g.set_priors(m, [g.Normal(0, 1)] * 3)
    assert len(g.get_priors(m.means[1])) == 2 and g.prior_logpdf(m) == pytest.approx(sum(stats.norm.logpdf([0.4, 1.0, -1.0])))


def test_fixed_kernel_exposes_the_free_slots_and_contributes_nothing():
    inner = g.SEArd([0.0, 0.1], 0.2)
    pri = [g.Normal(0, 1), g.Normal(0, 2), g.Normal(0, 3)]
    g.set_priors(inner, pri)
    fk = g.fix(inner, 1)  # parameter 1 frozen: free = [0, 2]
    assert g.get_priors(fk) == [pri[0], pri[2]]  # fixed_kernel.jl:78-84
    new = [g.Uniform(-1, 1), g.Uniform(-2, 2)]
    g.set_priors(fk, new)  # fixed_kernel.jl:86-90
    assert g.get_priors(inner) == [new[0], pri[1], new[1]]
    assert g.prior_logpdf(fk) == 0.0 and list(g.prior_gradlogpdf(fk)) == [0.0, 0.0]  # fixed_kernel.jl:92-98


def _bare_gpe(mean, kernel, log_noise):
    gp = object.__new__(gpe_mod.GPE)  # the constructor fits on the device; only the parameter objects are needed here
    gp.mean, gp.kernel, gp.logNoise, gp._noise_param = mean, kernel, log_noise, None
    return gp


def test_gpe_prior_terms_follow_the_parameter_switches():
    mean, kern = g.MeanConst(0.5), g.SEIso(0.2, -0.1)
    gp = _bare_gpe(mean, kern, -1.2)
    assert gp._prior_logpdf() == 0.0 and list(gp.prior_gradlogpdf()) == [0.0, 0.0, 0.0, 0.0]
    g.set_priors(gp.noise_param, [g.Normal(-1.0, 0.5)])
    g.set_priors(mean, [g.Normal(0.0, 2.0)])
    g.set_priors(kern, [g.Normal(0.0, 1.0), g.Normal(0.0, 1.0)])
    expect = (stats.norm(-1.0, 0.5).logpdf(-1.2) + stats.norm(0, 2).logpdf(0.5) + stats.norm.logpdf(0.2) + stats.norm.logpdf(-0.1))
    assert gp._prior_logpdf() == pytest.approx(expect, rel=1e-13)  # GPE.jl:348
    full = gp.prior_gradlogpdf()  # order: noise, mean, kernel (GPE.jl:514-526)
    np.testing.assert_allclose(full, [0.2 / 0.25, -0.5 / 4.0, -0.2, 0.1], rtol=1e-13)
    np.testing.assert_allclose(gp.prior_gradlogpdf(noise=False, domean=False), full[2:], rtol=1e-13)
    np.testing.assert_allclose(gp.prior_gradlogpdf(kern=False), full[:2], rtol=1e-13)
    gp.logNoise = -0.7  # the noise view reads the live value
    assert gp.prior_gradlogpdf()[0] == pytest.approx(-(-0.7 + 1.0) / 0.25)


def test_a_gp_is_not_kept_alive_by_its_noise_parameter_view():
    """gp.noise_param must not hold the GP strongly: a reference cycle would keep the device buffers of every model alive
    until the cyclic collector runs (the full GPU suite ran out of HBM that way)."""
    import gc
    import weakref

    class _GP:
        logNoise = -1.0

    gc.disable()
    try:
        gp = _GP()
        gp._noise_param = g.priors.NoiseParam(gp) if hasattr(g, "priors") else __import__("gpmi355x.priors", fromlist=["NoiseParam"]).NoiseParam(gp)
        assert gp._noise_param.get_params() == [-1.0]
        r = weakref.ref(gp)
        del gp
        assert r() is None  # freed by reference counting alone
    finally:
        gc.enable()
