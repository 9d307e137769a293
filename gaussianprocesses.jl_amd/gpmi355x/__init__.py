"""gpmi355x — MI355X-native exact-GP fit/predict behind the GaussianProcesses.jl API.

Host-side mirror of the reference's GP()/GPE/predict_f/optimize! surface and Kernel/Mean
hierarchy; all covariance / Cholesky / solve arithmetic runs in libgpmi.so (HIP, gfx950).
Importing this package never imports anything under oracle/ and there is no CPU fallback.
"""
from ._lib import ArgumentError, Context, DeviceError, PosDefException, load  # noqa: F401
from .gpe import (FITC, GP, GPE, HIPPDMat, get_params, logp_LOO, optimize, optimize_bounds, predict_f, predict_LOO, predict_y,  # noqa: F401
                  set_params,
                  update_mll, update_target)
from .kernels import (RQ, SE, Const, FixedKernel, Kernel, Masked, Mat12Ard, Mat12Iso, Mat32Ard,  # noqa: F401
                      Mat32Iso, Mat52Ard, Mat52Iso, Matern, Noise, ProdKernel, RQArd, RQIso, SEArd,
                      SEIso, SumKernel, fix, from_spec)
from .priors import Normal, Uniform, get_priors, prior_gradlogpdf, prior_logpdf, set_priors  # noqa: F401
from . import priors  # noqa: F401
from .means import Mean, MeanConst, MeanLin, MeanPeriodic, MeanPoly, MeanZero, ProdMean, SumMean  # noqa: F401
from .sparse import FullyIndepPDMat, FullyIndepStrat  # noqa: F401


def cov(kernel, X1, X2=None, dtype="float64", ctx=None):
    """cov(k, X1[, X2]) — src/kernels/kernels.jl:31-37,77 on the device."""
    import ctypes as C

    import numpy as np

    from . import _lib

    ctx = ctx or _lib.Context.default()
    bits = 64 if np.dtype(dtype) == np.float64 else 32
    dt = _lib.np_dtype(bits)
    a = _lib.colmajor(X1, dt)
    d, n1 = a.shape
    kd, keep = kernel.descriptor(d)
    if X2 is None:
        out = np.empty((n1, n1), dtype=dt, order="F")
        rc = _lib.load().gpmi_cov(ctx.h, C.byref(kd), bits, d, n1, a.ctypes.data, 0, None, out.ctypes.data)
    else:
        b = _lib.colmajor(X2, dt)
        if b.shape[0] != d:
            raise ArgumentError("X1 and X2 must have same dimension")
        out = np.empty((n1, b.shape[1]), dtype=dt, order="F")
        rc = _lib.load().gpmi_cov(ctx.h, C.byref(kd), bits, d, n1, a.ctypes.data, b.shape[1], b.ctypes.data, out.ctypes.data)
    del keep
    ctx.check(rc)
    return out
