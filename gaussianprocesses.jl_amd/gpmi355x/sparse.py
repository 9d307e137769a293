"""FITC on the MI355X path — the reference's `FullyIndepStrat` covariance strategy (src/sparse/).

Mirrors:
    FullyIndepStrat(inducing)                               fully_indep_train_conditional.jl:111-113
    FITC(x, inducing, y, mean, kernel, logNoise)            fully_indep_train_conditional.jl:333-336
    FullyIndepPDMat (the `gp.cK` of a FITC model)           :8-19   — here a device handle (gpmi_fitc)
    update_cK! / update_mll! / predict_f on that strategy   :134-156, src/GPE.jl:202-212, :321-329
`inducing` is d × m, one inducing point per column, like `x`.  SoR / DTC / FSA are outside this build.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class FullyIndepStrat:
    def __init__(self, inducing):
        inducing = np.asarray(inducing, dtype=np.float64)
        if inducing.ndim == 1:
            inducing = inducing[None, :]
        self.inducing = inducing


class FullyIndepPDMat:
    """Device-resident FITC covariance  Σ ≈ Kfu Kuu⁻¹ Kuf + Λ  (what alloc_cK(::FullyIndepStrat, nobs) returns)."""

    def __init__(self, ctx, x_colmajor, inducing_colmajor, bits):
        self.ctx, self.bits = ctx, bits
        d, n = x_colmajor.shape
        du, m = inducing_colmajor.shape
        if du != d:
            raise _lib.ArgumentError("inducing points and observations do not have consistent dimensions")
        self.dim, self.n, self.m = d, n, m
        h = C.c_void_p()
        ctx.check(_lib.load().gpmi_fitc_create(ctx.h, bits, d, n, x_colmajor.ctypes.data, m, inducing_colmajor.ctypes.data,
                                               C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.ctx.h:
                _lib.load().gpmi_fitc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def alpha_u(self):  # get_alpha_u, fully_indep_train_conditional.jl:279-286
        out = np.empty(self.m, dtype=_lib.np_dtype(self.bits))
        self.ctx.check(_lib.load().gpmi_fitc_alpha_u(self.h, out.ctypes.data))
        return out


def fitc_update_mll(gp):
    """update_mll!(gp) with gp.covstrat::FullyIndepStrat — one device pass (gpmi_fitc_fit)."""
    if np.ndim(gp.logNoise) != 0:
        raise _lib.ArgumentError("FITC takes a scalar logNoise (fully_indep_train_conditional.jl:134)")
    dt = _lib.np_dtype(gp.bits)
    ymu = np.ascontiguousarray(gp.y - gp.mean.mean(gp.x), dtype=dt)
    kd, keep = gp.kernel.descriptor(gp.dim)
    alpha = np.empty(gp.nobs, dtype=dt)
    mll, info = C.c_double(), C.c_int64()
    rc = _lib.load().gpmi_fitc_fit(gp.cK.h, C.byref(kd), float(gp.logNoise), ymu.ctypes.data, C.byref(mll), alpha.ctypes.data,
                                   C.byref(info))
    del keep
    gp.ctx.check(rc, info.value)
    gp.alpha, gp.mll = alpha, mll.value
    return gp


def fitc_predict_f(gp, xp, full_cov):
    dt = _lib.np_dtype(gp.bits)
    P = xp.shape[1]
    mx = np.ascontiguousarray(gp.mean.mean(xp), dtype=dt)
    mu = np.empty(P, dtype=dt)
    var = np.empty((P, P), dtype=dt, order="F") if full_cov else np.empty(P, dtype=dt)
    kd, keep = gp.kernel.descriptor(gp.dim)
    rc = _lib.load().gpmi_fitc_predict(gp.cK.h, C.byref(kd), P, xp.ctypes.data, mx.ctypes.data, 1 if full_cov else 0,
                                       mu.ctypes.data, var.ctypes.data)
    del keep
    gp.ctx.check(rc)
    return mu, var
