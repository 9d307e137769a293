"""The two-level factorisation and whitening (chol.h: super-panels of 512 / 1024 / 2048 columns, the next diagonal
super-block factored and inverted on the side stream under the K = W trailing update, rows below solved through the
explicit inverse of the super-block) at sizes the oracle finishes in seconds: the width thresholds are lowered through
the bring-up environment of a fresh context so that every width, every transition between widths, ragged last blocks,
the substitution fall-back and a pivot failing on the side stream are exercised.  Reference semantics:
make_posdef! / dpotrf info (src/GP.jl:101-112), update_mll! (src/GPE.jl:202-212), predict_f (src/GP.jl:64-79)."""
import math
import os

import numpy as np
import pytest

import gpmi355x as g
from oracle import gp_oracle as G

pytestmark = pytest.mark.gpu


def _ctx(monkeypatch, **env):
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return g.Context(0)  # the knobs are read once per context


def _close(a, b, rtol, atol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    assert np.all(err <= atol + rtol * np.abs(b)), f"{what}: max abs err {err.max():.3e}"


SPEC = ("sum", ("se_ard", [math.log(0.4), math.log(0.5), math.log(0.6), math.log(0.7)], 0.0), ("mat32_iso", math.log(0.8), -0.5))


def _check(ctx, n, p, ln=math.log(0.15), noise_vec=False):
    x, y, xs = G.synthetic_inputs(n, 4, p=p)
    log_noise = ln + 0.3 * np.sin(np.arange(n)) if noise_vec else ln
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(SPEC), log_noise, ctx=ctx)
    ref = G.update_mll(SPEC, x, y, log_noise)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-9)
    assert gp.cK.logdet() == pytest.approx(ref["logdet"], rel=1e-10)
    _close(gp.alpha, ref["alpha"], 1e-6, 1e-8 * np.abs(ref["alpha"]).max(), "alpha")
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(SPEC, x, ref, xs)
    _close(mu, mu_o, 1e-6, 1e-8, "mu")
    _close(s2, s2_o, 1e-5, 1e-9, "sigma2")
    return gp, ref


@pytest.mark.parametrize("n,sup,la_min", [
    (2900, "512,1024,2048", 256),    # 2048 + 512 + 256 + a ragged 128-column block, look-ahead everywhere
    (2900, "512,1024,2048", 100000), # the same widths, serial order (no look-ahead)
    (4100, "600,1500,3000", 256),    # 2048, 1024, 512, 512, then the plain NB = 256 tail
    (3333, "0,1024,0", 256),         # only 1024-wide super-panels; n not a multiple of 64
    (1600, "512,0,0", 256),          # 512-wide only
])
def test_every_width_and_transition_vs_oracle(monkeypatch, n, sup, la_min):
    ctx = _ctx(monkeypatch, GPMI_SUPER=sup, GPMI_LOOKAHEAD_MIN=la_min)
    _check(ctx, n, 300)


def test_substitution_fallback_matches_the_inverse_path(monkeypatch):
    """GPMI_SUPER_INV=0: rows below a super-panel by NB-block substitution (the form nugget-regularised matrices use)."""
    a = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256, GPMI_SUPER_INV=0)
    gp_a, _ = _check(a, 3000, 64)
    b = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256, GPMI_SUPER_INV=1)
    gp_b, _ = _check(b, 3000, 64)
    assert gp_a.mll == pytest.approx(gp_b.mll, rel=1e-12)


def test_vector_noise_and_the_refined_substitution_for_tiny_noise(monkeypatch):
    ctx = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256)
    _check(ctx, 2700, 100, noise_vec=True)
    # sigma^2 = e^-14 < 1e-5 k(x,x): refine_solves, i.e. no explicit super-block inverse (api.hip fit_t)
    x, y, _ = G.synthetic_inputs(2300, 4, p=4)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(SPEC), -7.0, ctx=ctx)
    ref = G.update_mll(SPEC, x, y, -7.0)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-7)


@pytest.mark.parametrize("pivot", [700, 1301, 2600])
def test_failing_pivot_inside_a_super_block_on_the_side_stream(monkeypatch, pivot):
    """dpotrf's info (src/GP.jl:110): the first non-positive pivot, 1-based — here inside the first super-block (main
    stream), inside the second (factored on the side stream under the first K = 1024 update) and inside the third."""
    ctx = _ctx(monkeypatch, GPMI_SUPER="0,1024,0", GPMI_LOOKAHEAD_MIN=256)
    n = 3300
    x = np.arange(n, dtype=np.float64)[None, :]
    x[0, pivot - 1] = x[0, 40]  # a duplicated point: with no noise the pivot of the second copy is exactly <= 0
    y = np.random.default_rng(1).standard_normal(n)
    with pytest.raises(g.PosDefException) as ei:
        g.GP(x, y, g.MeanZero(), g.SEIso(-3.0, 0.0), -400.0, ctx=ctx)
    assert ei.value.info == pivot
    gp = g.GP(x, y, g.MeanZero(), g.SEIso(-3.0, 0.0), -1.0, ctx=ctx)  # the context is still usable afterwards
    assert np.isfinite(gp.mll)


def test_gradient_and_loo_through_the_two_level_whitening(monkeypatch):
    ctx = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256, GPMI_WHITEN_SUPER=512)
    x, y, _ = G.synthetic_inputs(1900, 4, p=4)
    ln = math.log(0.15)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(SPEC), ln, ctx=ctx)
    ref = G.update_mll(SPEC, x, y, ln)
    gp.update_dmll()
    d_o = G.update_dmll(SPEC, x, y, ln, fit=ref)["dmll"]
    _close(gp.dmll, d_o, 1e-7, 1e-8 * np.abs(d_o).max(), "dmll")
    mu, s2 = gp.predict_LOO()
    mu_o, s2_o = G.predict_loo(ref, y)
    _close(s2, s2_o, 1e-7, 1e-12, "loo variance")
    _close(mu, mu_o, 1e-7, 1e-9, "loo mean")


def test_fp32_two_level_vs_fp64_oracle(monkeypatch):
    ctx = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256)
    x, y, xs = G.synthetic_inputs(3100, 4, p=50)
    gp = g.GP(x.astype(np.float32), y, g.MeanZero(), g.from_spec(SPEC), math.log(0.15), dtype=np.float32, ctx=ctx)
    ref = G.update_mll(SPEC, x, y, math.log(0.15))
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-2)
    mu, s2 = gp.predict_f(xs.astype(np.float32))
    mu_o, s2_o = G.predict_f(SPEC, x, ref, xs)
    _close(mu, mu_o, 1e-2, 1e-3, "mu fp32")
    _close(s2, s2_o, 1e-2, 1e-3, "sigma2 fp32")


def test_whitening_through_nb_blocks_matches_the_stored_super_inverses(monkeypatch):
    """GPMI_WHITEN_INV=0: predict_f whitens through the 256-block inverses although the fit kept its super-block inverses."""
    a = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256, GPMI_WHITEN_INV=0)
    gp_a, _ = _check(a, 3000, 200)
    b = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256, GPMI_WHITEN_INV=1)
    gp_b, _ = _check(b, 3000, 200)
    xs = G.synthetic_inputs(3000, 4, p=200)[2]
    np.testing.assert_allclose(gp_a.predict_f(xs)[1], gp_b.predict_f(xs)[1], rtol=1e-9, atol=1e-12)



@pytest.mark.parametrize("n,dtype,cumask", [(2500, np.float64, 0), (4100, np.float64, 0), (6000, np.float64, 0), (6000, np.float64, 1),
                                            (5000, np.float32, 0)])
def test_update_in_256x128_tiles_small(monkeypatch, n, dtype, cumask):
    """The trailing update in 256 x 128 tiles (csrc/update256.hip) at sizes the oracle factors in seconds: GPMI_UPDATE256_MIN=1
    sends EVERY eligible update of the dense factorisation through it (by default only launches of >= 1024 tiles take it), so edge
    tiles in both directions, the carried right-hand-side row, launches smaller than the grid and the queue all run; K = 256 in
    fp32 is 8 slabs (not eligible) and stays on the 128 x 128 kernel.  The kernel is not used on the CU-masked update stream, which
    is where look-ahead updates of factorisations under 32 768 rows go: GPMI_CUMASK=0 (free slots everywhere) puts the look-ahead
    updates — the lower region with a row offset, grid of 248 workgroups — through it as well; cumask = 1 keeps the default (only
    the serial-order updates take it).  mll / alpha / predictions against the oracle."""
    monkeypatch.setenv("GPMI_UPDATE256_MIN", "1")
    if not cumask:
        monkeypatch.setenv("GPMI_CUMASK", "0")
    monkeypatch.setenv("GPMI_SUPER", "1024,2048,100000")  # super-panels of 512 from 1024 rows on, 1024 from 2048: K = 512 / 1024 updates at these sizes
    ctx = g.Context(0)
    d = 5
    x, y, xs = G.synthetic_inputs(n, d, p=64)
    ll = [math.log(0.6)] * d
    spec = ("se_ard", ll, 0.0)
    gp = g.GP(x.astype(dtype), y, g.MeanZero(), g.from_spec(spec), math.log(0.1), dtype=dtype, ctx=ctx)
    ref = G.update_mll(spec, x, y, math.log(0.1))
    tol = 1e-9 if dtype == np.float64 else 2e-3
    assert abs(gp.mll - ref["mll"]) <= tol * abs(ref["mll"]), (gp.mll, ref["mll"])
    np.testing.assert_allclose(gp.alpha, ref["alpha"], rtol=0, atol=(1e-7 if dtype == np.float64 else 5e-2) * np.abs(ref["alpha"]).max())
    mu, var = gp.predict_f(xs.astype(dtype))
    rmu, rvar = G.predict_f(spec, x, ref, xs)
    np.testing.assert_allclose(mu, rmu, rtol=0, atol=(1e-8 if dtype == np.float64 else 2e-3) * np.abs(rmu).max())
    np.testing.assert_allclose(var, rvar, rtol=0, atol=(1e-8 if dtype == np.float64 else 2e-3))


@pytest.mark.parametrize("n,dtype", [(4100, np.float64), (6000, np.float64), (5000, np.float32)])
def test_panel_solves_in_256x128_tiles(monkeypatch, n, dtype):
    """Round 6: the panel solve X LW' (GEMM_OVERWRITE | GEMM_KEND_COL: a column tile's K loop ends at its last column — LW is the explicit
    lower-triangular inverse of the diagonal super-block, chol.h rows_below_super) in 256 x 128 tiles (update256_kernel<T, 0, true, false,
    true>).  By default only tall solves (>= 8192 rows, >= 1024 tiles) take it; GPMI_UPDATE256_RECT=1 / _MIN=1 send EVERY solve through an
    inverse — and predict_f's V = R LW' with its 64 test rows — through it: edge tiles in both directions, the carried row, K loops of
    8 ... 64 slabs.  Against the oracle, and against the same fit with GPMI_UPDATE256_KEND=0 (the 128 x 128 kernel of rounds 1-5)."""
    monkeypatch.setenv("GPMI_UPDATE256_MIN", "1")
    monkeypatch.setenv("GPMI_UPDATE256_RECT", "1")
    monkeypatch.setenv("GPMI_CUMASK", "0")
    monkeypatch.setenv("GPMI_SUPER", "1024,2048,100000")
    d = 5
    x, y, xs = G.synthetic_inputs(n, d, p=64)
    spec = ("se_ard", [math.log(0.6)] * d, 0.0)
    out = {}
    for kend in ("1", "0"):
        monkeypatch.setenv("GPMI_UPDATE256_KEND", kend)
        ctx = g.Context(0)
        gp = g.GP(x.astype(dtype), y, g.MeanZero(), g.from_spec(spec), math.log(0.1), dtype=dtype, ctx=ctx)
        mu, var = gp.predict_f(xs.astype(dtype))
        out[kend] = (gp.mll, np.array(gp.alpha, dtype=np.float64), np.array(mu, dtype=np.float64), np.array(var, dtype=np.float64),
                     np.array(gp.cK.factor_diag(), dtype=np.float64))
        del gp
        ctx.close()
    ref = G.update_mll(spec, x, y, math.log(0.1))
    rmu, rvar = G.predict_f(spec, x, ref, xs)
    f64 = dtype == np.float64
    mll, alpha, mu, var, dg = out["1"]
    assert abs(mll - ref["mll"]) <= (1e-9 if f64 else 2e-3) * abs(ref["mll"]), (mll, ref["mll"])
    np.testing.assert_allclose(alpha, ref["alpha"], rtol=0, atol=(1e-7 if f64 else 5e-2) * np.abs(ref["alpha"]).max())
    np.testing.assert_allclose(mu, rmu, rtol=0, atol=(1e-8 if f64 else 2e-3) * np.abs(rmu).max())
    np.testing.assert_allclose(var, rvar, rtol=0, atol=(1e-8 if f64 else 2e-3))
    # the two kernels accumulate a tile's K loop in the same slab order: rounding-level agreement of everything downstream
    tol = 1e-11 if f64 else 1e-4
    assert abs(mll - out["0"][0]) <= tol * abs(mll)
    np.testing.assert_allclose(dg, out["0"][4], rtol=tol)
    np.testing.assert_allclose(mu, out["0"][2], rtol=0, atol=tol * 10 * np.abs(rmu).max())


def test_update_in_256x128_tiles_atomic_epilogue(monkeypatch):
    """GPMI_UPDATE256_ATOMIC=1: the C tile of a subtracting launch leaves as no-return global_atomic_add (one add per element: the same
    numbers as load / add / store).  Same factorisation with and without: mll, alpha and the factor's diagonal agree to rounding."""
    x, y, _ = G.synthetic_inputs(6000, 5, p=4)
    kern = lambda: g.SEArd([math.log(0.6)] * 5, 0.0)
    monkeypatch.setenv("GPMI_UPDATE256_MIN", "1")
    monkeypatch.setenv("GPMI_CUMASK", "0")
    monkeypatch.setenv("GPMI_SUPER", "1024,2048,100000")
    out = []
    for atom in ("1", "0"):
        monkeypatch.setenv("GPMI_UPDATE256_ATOMIC", atom)
        gp = g.GP(x, y, g.MeanZero(), kern(), math.log(0.1), ctx=g.Context(0))
        out.append((gp.mll, np.array(gp.alpha), np.array(gp.cK.factor_diag())))
    ref = G.update_mll(("se_ard", [math.log(0.6)] * 5, 0.0), x, y, math.log(0.1))
    assert abs(out[0][0] - ref["mll"]) <= 1e-9 * abs(ref["mll"])
    assert abs(out[0][0] - out[1][0]) <= 1e-12 * abs(out[1][0])
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=1e-10 * np.abs(out[1][1]).max())
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-12)


def test_update_in_256x128_tiles_matches_the_128_kernel_n12000(monkeypatch):
    """Same factorisation through both update kernels at a size where the look-ahead is active: the factor's
    diagonal, alpha and mll agree to rounding; a failing pivot is reported identically."""
    x, y, _ = G.synthetic_inputs(12000, 6, p=4)
    kern = lambda: g.SEArd([math.log(0.5)] * 6, 0.0)
    monkeypatch.setenv("GPMI_UPDATE256_MIN", "1")
    monkeypatch.setenv("GPMI_CUMASK", "0")  # free slots: the look-ahead updates go through the 256 x 128 kernel too
    a = g.GP(x, y, g.MeanZero(), kern(), math.log(0.05), ctx=g.Context(0))
    monkeypatch.setenv("GPMI_UPDATE256", "0")
    b = g.GP(x, y, g.MeanZero(), kern(), math.log(0.05), ctx=g.Context(0))
    assert abs(a.mll - b.mll) <= 1e-11 * abs(b.mll)
    np.testing.assert_allclose(a.alpha, b.alpha, rtol=0, atol=1e-9 * np.abs(b.alpha).max())
    np.testing.assert_allclose(a.cK.factor_diag(), b.cK.factor_diag(), rtol=1e-11)
    # a matrix that stops being positive definite late: both report the same pivot
    xd = x.copy()
    xd[:, 11000] = xd[:, 3]
    monkeypatch.delenv("GPMI_UPDATE256")
    errs = []
    for env in ("1", "0"):
        monkeypatch.setenv("GPMI_UPDATE256", env)
        with pytest.raises(g.PosDefException) as ei:
            g.GP(xd, y, g.MeanZero(), g.SEArd([math.log(0.5)] * 6, 0.0), -40.0, ctx=g.Context(0))
        errs.append(ei.value.info)
    assert errs[0] == errs[1] and errs[0] > 0
