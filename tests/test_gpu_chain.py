"""The persistent chain kernel (csrc/chain.hip): ONE launch per diagonal super-block — Cholesky factor, 64 x 64 diagonal inverses and the
explicit inverse of the block by a dataflow of 64 x 64 tile tasks — against LAPACK on the oracle's matrix and against the multi-launch
chain of rounds 1-4 (GPMI_CHAIN=0), for every block width the factorisation uses (256 tail blocks without inverse, 512, 1024, 2048),
with look-ahead (the launch runs beside the trailing update on the reserved compute units / free slots) and without, with 1, 3 and many
workgroups (any number must give the SAME bits: every tile is accumulated in a fixed order by whoever takes it), in fp32, on a blocked
handle (gpmi_dev super_factor), and with a pivot failing inside the launch.
Reference semantics: make_posdef! / dpotrf info (src/GP.jl:101-112), update_mll! (src/GPE.jl:202-212), predict_f (src/GP.jl:64-79)."""
import math

import numpy as np
import pytest

import gpmi355x as g
from oracle import gp_oracle as G

pytestmark = pytest.mark.gpu

SPEC = ("sum", ("se_ard", [math.log(0.4), math.log(0.5), math.log(0.6), math.log(0.7)], 0.0), ("mat32_iso", math.log(0.8), -0.5))
LN = math.log(0.15)


def _ctx(monkeypatch, **env):
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return g.Context(0)  # the knobs are read once per context


def _fit(ctx, n, dtype=np.float64, p=200):
    x, y, xs = G.synthetic_inputs(n, 4, p=p)
    gp = g.GP(x, y, g.MeanZero(), g.from_spec(SPEC), LN, ctx=ctx, dtype=dtype)
    return gp, x, y, xs


@pytest.mark.parametrize("n,sup,la_min", [
    (2900, "512,1024,2048", 256),     # a 2048 block, then 512, then 256-wide tail blocks (factor only) and a ragged last one; look-ahead
    (2900, "512,1024,2048", 100000),  # the same, serial order: the chain launches have the device to themselves (many workgroups)
    (4100, "600,1500,3000", 256),     # 2048, 1024, 512, 512, then the NB = 256 tail
    (3333, "0,1024,0", 256),          # 1024-wide blocks only; n not a multiple of 64
    (1600, "512,0,0", 256),           # 512-wide only
    (700, "512,1024,2048", 256),      # below every threshold: plain 256 panels
])
def test_chain_kernel_factor_inverse_and_solves_vs_lapack(monkeypatch, n, sup, la_min):
    ctx = _ctx(monkeypatch, GPMI_SUPER=sup, GPMI_LOOKAHEAD_MIN=la_min, GPMI_CHAIN=1)
    gp, x, y, xs = _fit(ctx, n)
    ref = G.update_mll(SPEC, x, y, LN)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-10)
    U = np.triu(ref["U"])
    np.testing.assert_allclose(np.triu(gp.cK.cholfactors()), U, rtol=1e-9, atol=1e-12)            # the factor itself, every tile
    np.testing.assert_allclose(gp.alpha, ref["alpha"], rtol=1e-6, atol=1e-8 * np.abs(ref["alpha"]).max())
    mu, s2 = gp.predict_f(xs)                                                                      # whitening through the stored block inverses
    mu_o, s2_o = G.predict_f(SPEC, x, ref, xs)
    np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(s2, s2_o, rtol=1e-5, atol=1e-9)
    Kinv_diag = np.diag(np.linalg.inv(U.T @ U))
    np.testing.assert_allclose(gp.cK.inv_diag(), Kinv_diag, rtol=1e-6)                             # L^-T rows through the same inverses


@pytest.mark.parametrize("n,sup", [(2900, "512,1024,2048"), (3333, "0,1024,0")])
def test_any_number_of_workgroups_gives_the_same_bits_and_matches_the_multi_launch_chain(monkeypatch, n, sup):
    facs = {}
    for tag, env in (("default", {}), ("one", {"GPMI_CHAIN_WGS": 1}), ("three", {"GPMI_CHAIN_WGS": 3}), ("many", {"GPMI_CHAIN_WGS": 61}),
                     ("old", {"GPMI_CHAIN": 0})):
        monkeypatch.delenv("GPMI_CHAIN_WGS", raising=False)
        monkeypatch.delenv("GPMI_CHAIN", raising=False)
        ctx = _ctx(monkeypatch, GPMI_SUPER=sup, GPMI_LOOKAHEAD_MIN=256, **env)
        gp, x, y, xs = _fit(ctx, n)
        facs[tag] = (np.triu(gp.cK.cholfactors()), gp.mll, np.array(gp.alpha))
        del gp
        ctx.close()
    for tag in ("one", "three", "many"):
        assert np.array_equal(facs[tag][0], facs["default"][0]), tag       # a dataflow, not a race: same tiles, same order, same bits
        assert facs[tag][1] == facs["default"][1]
    np.testing.assert_allclose(facs["default"][0], facs["old"][0], rtol=1e-10, atol=1e-13)    # rounding-level differences only
    assert facs["default"][1] == pytest.approx(facs["old"][1], rel=1e-12)


def test_chain_stress_many_workgroup_counts(monkeypatch):
    """The canary for the inter-workgroup visibility protocol (chain.hip publish() / wait_flag: sc1 write-through stores, drain, flag; relaxed
    poll, compiler barrier, sc1 loads — no fences): a toolchain or runtime that breaks it shows up as a factor whose bits depend on who
    computed which tile.  Ten workgroup counts x repeated fits with refreshed parameters, every factor bit-identical to the one-workgroup
    (purely sequential, nothing to synchronise) run of the same parameters."""
    n, sup = 4100, "600,1500,3000"
    x, y, _ = G.synthetic_inputs(n, 4, p=8)
    lns = [LN + 0.01 * r for r in range(4)]

    def run(wgs):
        monkeypatch.delenv("GPMI_CHAIN_WGS", raising=False)
        env = {} if wgs is None else {"GPMI_CHAIN_WGS": wgs}
        ctx = _ctx(monkeypatch, GPMI_SUPER=sup, GPMI_LOOKAHEAD_MIN=256, **env)
        gp = g.GP(x, y, g.MeanZero(), g.from_spec(SPEC), LN, ctx=ctx)
        out = []
        for ln in lns:
            gp.set_params(np.concatenate([[ln], gp.get_params()[1:]]))
            gp.update_mll()
            out.append((gp.mll, np.triu(gp.cK.cholfactors())))
        del gp
        ctx.close()
        return out

    base = run(1)
    for wgs in (None, 2, 5, 8, 13, 16, 24, 32, 48, 64):
        got = run(wgs)
        for (m0, f0), (m1, f1) in zip(base, got):
            assert m0 == m1, wgs
            assert np.array_equal(f0, f1), wgs


def test_chain_kernel_fp32(monkeypatch):
    ctx = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256)
    gp, x, y, xs = _fit(ctx, 2900, dtype=np.float32)
    x64 = np.asarray(gp.x, dtype=np.float64)
    ref = G.update_mll(SPEC, x64, y, LN)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-2)                                              # north_star's fp32 bar
    mu, s2 = gp.predict_f(xs)
    mu_o, s2_o = G.predict_f(SPEC, x64, ref, np.asarray(xs, dtype=np.float32).astype(np.float64))
    np.testing.assert_allclose(mu, mu_o, rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(s2, s2_o, rtol=1e-2, atol=1e-3)
    U32 = np.triu(gp.cK.cholfactors()).astype(np.float64)
    assert np.abs(U32.T @ U32 - np.triu(ref["U"]).T @ np.triu(ref["U"])).max() <= 2e-4               # K + s2 I reconstructed from the fp32 factor


@pytest.mark.parametrize("bad", [70, 600, 1500, 2300])
def test_pivot_failing_inside_the_chain_launch(monkeypatch, bad):
    """PosDefException(info): points far apart (K ~ I) with ONE exact duplicate and no noise to speak of — the Cholesky stops at the duplicate's
    column, somewhere inside the first 2048 block, with dpotrf's own 1-based pivot; the context (and the chain's synchronisation area) factors a
    good matrix right after (as test_not_posdef_late_pivot / _during_lookahead do for the multi-launch chain)."""
    ctx = _ctx(monkeypatch, GPMI_SUPER="512,1024,2048", GPMI_LOOKAHEAD_MIN=256)
    n = 2900
    x = np.arange(n, dtype=np.float64)[None, :]
    x[0, bad] = x[0, 17]
    y = np.random.default_rng(0).standard_normal(n)
    with pytest.raises(g.PosDefException) as ei:
        g.GP(x, y, g.MeanZero(), g.SEIso(-3.0, 0.0), -400.0, ctx=ctx)
    assert ei.value.info == bad + 1
    gp, x2, y2, xs = _fit(ctx, n)
    ref = G.update_mll(SPEC, x2, y2, LN)
    assert gp.mll == pytest.approx(ref["mll"], rel=1e-10)


def test_chain_kernel_on_a_blocked_handle(monkeypatch):
    """gpmi_gp_create_blocked: the blocked driver's super_factor is the same launch (explicit inverse with leading dimension w)"""
    from gpmi355x import dist as gd

    ctx = _ctx(monkeypatch, GPMI_CHAIN=1)
    x, y, xs = G.synthetic_inputs(5000, 4, p=64)
    for block in (512, 1024, 2048):
        gp = gd.ShardedGPE(x, y, g.MeanZero(), g.from_spec(SPEC), LN, ctx=ctx, block=block)
        ref = G.update_mll(SPEC, x, y, LN)
        assert gp.mll == pytest.approx(ref["mll"], rel=1e-10)
        np.testing.assert_allclose(gp.cK.factor_diag(), np.diag(ref["U"]), rtol=1e-9)
        mu, s2 = gp.predict_f(xs)
        mu_o, s2_o = G.predict_f(SPEC, x, ref, xs)
        np.testing.assert_allclose(mu, mu_o, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(s2, s2_o, rtol=1e-5, atol=1e-9)
        del gp

