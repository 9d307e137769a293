"""optimize!'s host logic (src/optimize.jl:19-83, src/GPE.jl:467-490) against a stand-in GP object: the assembly of the
box from noisebounds / meanbounds / kernbounds, the bound-constrained run, and the error contract (a PosDefException
during an evaluation restores the parameters and reports the point as infeasible).  CPU only: the device supplies
target and gradient in the real object (tests/test_gpu_parity.py::test_optimize_with_device_gradient_...)."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussianprocesses.jl_amd"))
import gpmi355x as g  # noqa: E402


class _Params:
    def __init__(self, v):
        self.v = list(v)

    def get_params(self):
        return list(self.v)

    def set_params(self, hyp):
        self.v = list(hyp)

    def num_params(self):
        return len(self.v)


class StandInGP:
    """target = -|theta - centre|^2 over (logNoise, mean params, kernel params); not positive definite left of `wall`."""

    def __init__(self, centre, wall=-math.inf):
        self.logNoise = 0.0
        self.mean = _Params([0.0, 0.0])
        self.kernel = _Params([0.0, 0.0, 0.0])
        self.centre = np.asarray(centre, dtype=float)
        self.wall = wall
        self.evals = 0

    def get_params(self, noise=True, domean=True, kern=True):
        return ([self.logNoise] if noise else []) + (self.mean.get_params() if domean else []) + (self.kernel.get_params() if kern else [])

    def num_params(self, **kw):
        return len(self.get_params(**kw))

    def set_params(self, hyp, noise=True, domean=True, kern=True):
        hyp = list(hyp)
        if noise:
            self.logNoise = hyp.pop(0)
        if domean:
            self.mean.set_params([hyp.pop(0) for _ in range(2)])
        if kern:
            self.kernel.set_params([hyp.pop(0) for _ in range(3)])

    def _full(self):
        return np.array(self.get_params())

    def update_target_and_dtarget(self, noise=True, domean=True, kern=True):
        self.evals += 1
        th = self._full()
        if th[0] < self.wall:
            raise g.PosDefException(3)
        self.target = -float(np.sum((th - self.centre) ** 2))
        mask = np.array([noise] + [domean] * 2 + [kern] * 3)
        self.dtarget = (-2.0 * (th - self.centre))[mask]

    def update_target(self):
        th = self._full()
        self.target = -float(np.sum((th - self.centre) ** 2))


def test_bounds_follow_get_params_order_and_default_to_infinity():
    gp = StandInGP(np.zeros(6))
    assert g.optimize_bounds(gp) is None  # no pair at all: the reference runs the unconstrained optimizer
    box = g.optimize_bounds(gp, kernbounds=([-1.0, -2.0, -3.0], [1.0, 2.0, 3.0]))
    assert box == [(-math.inf, math.inf)] * 3 + [(-1.0, 1.0), (-2.0, 2.0), (-3.0, 3.0)]
    box = g.optimize_bounds(gp, noisebounds=([-4.0], [0.5]), meanbounds=([0.0, 0.0], [1.0, 1.0]), kern=False)
    assert box == [(-4.0, 0.5), (0.0, 1.0), (0.0, 1.0)]
    box = g.optimize_bounds(gp, noisebounds=(-4.0, 0.5), noise=True, domean=False, kern=False)  # scalars are accepted
    assert box == [(-4.0, 0.5)]
    with pytest.raises(g.ArgumentError):
        g.optimize_bounds(gp, kernbounds=([0.0], [1.0]))


def test_bound_constrained_run_stops_at_the_box():
    centre = np.array([-3.0, 0.5, -0.5, 2.0, -2.0, 0.25])
    gp = StandInGP(centre)
    res = g.optimize(gp, options={"maxiter": 200}, noisebounds=([-1.0], [1.0]),
                     kernbounds=([-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]))
    expect = np.array([-1.0, 0.5, -0.5, 1.0, -1.0, 0.25])  # clipped where the box cuts, free elsewhere (mean unbounded)
    np.testing.assert_allclose(res.x, expect, atol=1e-6)
    np.testing.assert_allclose(gp.get_params(), expect, atol=1e-6)  # set_params!(minimizer) + update_target!
    assert gp.target == pytest.approx(-float(np.sum((expect - centre) ** 2)), abs=1e-8)


def test_switches_select_the_optimised_block():
    centre = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    gp = StandInGP(centre)
    g.optimize(gp, noise=False, domean=False, options={"maxiter": 100})
    np.testing.assert_allclose(gp.get_params(), [0.0, 0.0, 0.0, 4.0, 5.0, 6.0], atol=1e-6)


def test_posdef_failure_is_an_infeasible_point_and_parameters_are_restored():
    centre = np.array([-2.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    gp = StandInGP(centre, wall=-1.0)  # the optimum lies in the region where the factorisation fails
    res = g.optimize(gp, options={"maxiter": 50})
    assert math.isfinite(gp.target) and gp.get_params()[0] >= -1.0  # never left in the infeasible region
    assert res.x[0] >= -1.0 and res.x[0] < 0.0  # moved towards the optimum, stopped in front of the wall


class _Unsupported(StandInGP):
    """a model whose gradient the device does not cover: every evaluation raises ArgumentError, whatever the parameters"""

    def update_target_and_dtarget(self, **kw):
        self.evals += 1
        raise g.ArgumentError("update_dmll: not supported for this model")


def test_capability_errors_surface_instead_of_a_converged_result():
    gp = _Unsupported(np.zeros(6))
    with pytest.raises(g.ArgumentError, match="not supported"):
        g.optimize(gp)
    assert gp.evals == 1  # raised on the starting point, before the optimiser ran


def test_start_that_is_not_positive_definite_raises():
    gp = StandInGP(np.zeros(6), wall=1.0)  # logNoise = 0 lies left of the wall
    with pytest.raises(g.PosDefException):
        g.optimize(gp)
