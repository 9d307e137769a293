"""Host logic (no GPU): the product's kernel serialiser agrees with the oracle's independent
flattening; parameter vectors follow the reference's ordering and transforms."""
import math

import numpy as np
import pytest

import gpmi355x as g
from kernel_cases import ALL, D, ids
from oracle import c_oracle
from oracle import gp_oracle as G


@pytest.mark.parametrize("spec", ALL, ids=ids(ALL))
def test_descriptor_matches_oracle_flatten(spec):
    k = g.from_spec(spec)
    ops, dims_off, dims, params = k.flat(D)
    o_ops, o_off, o_dims, o_par = c_oracle.flatten(spec, D)
    assert ops == list(o_ops)
    assert dims_off == list(o_off)
    assert dims == (list(o_dims) if dims else [])
    np.testing.assert_allclose(params, o_par, rtol=0, atol=0)
    if spec[0] != "fixed":
        assert k.num_params() == G.num_params(spec)


def test_param_roundtrip_and_ordering():
    # se_ard.jl:33-41  [ll..., lσ];  rq_ard.jl:36-45 [ll..., lσ, lα];  pair_kernel.jl:14-24 left then right
    k = g.SEArd([0.1, 0.2, 0.3], 0.4) + g.RQIso(0.5, 0.6, 0.7) * g.Noise(-1.0)
    np.testing.assert_allclose(k.get_params(), [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, -1.0], atol=1e-15)
    k.set_params([1, 2, 3, 4, 5, 6, 7, 8])
    np.testing.assert_allclose(k.get_params(), [1, 2, 3, 4, 5, 6, 7, 8], rtol=1e-14)
    assert k.kleft.il2[0] == pytest.approx(math.exp(-2.0))
    assert k.kright.kleft.alpha == pytest.approx(math.exp(7.0))
    with pytest.raises(g.ArgumentError):
        g.SEIso(0.0, 0.0).set_params([1.0])
    with pytest.raises(g.ArgumentError):
        g.SEArd([0.0, 0.0], 0.0).flat(3)  # wrong number of length scales for d = 3


def test_descriptor_has_no_dimension_or_parameter_limit():
    """round 4: the library sizes its kernel-program tables from the descriptor (include/gpmi.h: only the number of tree nodes is
    bounded) — the serialiser must not cap d or the parameter count either: d = 300 ARD leaves, a Masked leaf over 150 of them"""
    d = 300
    k = g.SEArd([0.01 * i for i in range(d)], 0.2) + g.Masked(g.Mat32Ard([0.1] * 150, 0.0), list(range(0, d, 2)))
    kd, keep = k.descriptor(d)
    assert kd.n_ops == 3 and kd.n_params == (d + 1) + (150 + 1) and k.num_params() == kd.n_params
    ops, dims_off, dims, params = k.flat(d)
    assert len(dims) == 150 and dims_off[-1] == 150 and len(params) == kd.n_params
    o = c_oracle.flatten(("sum", ("se_ard", [0.01 * i for i in range(d)], 0.2), ("masked", ("mat32_ard", [0.1] * 150, 0.0), list(range(0, d, 2)))), d)
    np.testing.assert_allclose(params, o[3], rtol=0, atol=0)
    del keep


def test_shortcut_constructors():
    # test/kernels.jl:184-205
    assert type(g.SE(0.0, 0.0)) is g.SEIso and type(g.SE([0.0, 1.0], 0.0)) is g.SEArd
    assert type(g.Matern(2.5, 0.0, 0.0)) is g.Mat52Iso and type(g.Matern(0.5, [0.0], 0.0)) is g.Mat12Ard
    assert type(g.RQ(0.0, 0.0, 0.0)) is g.RQIso
    with pytest.raises(g.ArgumentError):
        g.Matern(1.0, 0.0, 0.0)


def test_fixed_kernel_params():
    k = g.fix(g.SEIso(0.3, 0.5), 1)  # freeze lσ
    assert k.get_params() == pytest.approx([0.3])
    k.set_params([0.9])
    assert k.kernel.get_params() == pytest.approx([0.9, 0.5])


def test_golden_fixture_matches_oracle():
    """The committed fixture still equals what the oracle computes (fixture drift guard)."""
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_golden

    z = np.load(os.path.join(here, "golden", "simdata_bench_kernels.npz"))
    for i, name in enumerate(z["names"]):
        spec = make_golden.KERNS[str(name)]
        fit = G.update_mll(spec, z["x"], z["y"], float(z["log_noise"]), ("const", float(z["mean_const"])))
        assert fit["mll"] == pytest.approx(float(z[f"mll_{i}"]), rel=1e-12)
        np.testing.assert_allclose(fit["alpha"], z[f"alpha_{i}"], rtol=1e-9, atol=1e-12)
