"""GPE / GP() / update_mll! / predict_f — the reference's exact-GP API surface
(src/GPE.jl, src/GP.jl) driving the MI355X path through the C ABI.

Mirrors, with the same argument meaning and error behaviour:
    GP(x, y, mean, kernel, logNoise=-2.0)      src/GPE.jl:92-120
    GPE.fit!(gp, x, y)                         src/GPE.jl:128-138
    update_cK! / update_mll!(gp; noise, domean, kern)   src/GPE.jl:169-212
    initialise_target! / update_target!        src/GPE.jl:346-365 (target = mll + log prior, priors.py)
    predict_f / predict_y (full_cov)           src/GP.jl:64-84, src/GPE.jl:408-416
    get_params / set_params! / num_params      src/GPE.jl:447-512
    optimize!                                  src/optimize.jl:19-97 (error contract only)
The covariance strategy (src/GP.jl:10-20) is what this module replaces: `HIPPDMat`
plays the AbstractPDMat role of `gp.cK` (`\\`, whiten!, logdet, cholfactors).

Python has no `!`: update_mll!(gp) is `update_mll(gp)` etc.
x is d × N (one observation per column) exactly as in the reference.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from . import priors as _priors
from .kernels import Kernel
from .means import Mean, MeanZero


class HIPPDMat:
    """Device-resident (K + σ²I) = UᵀU.  AbstractPDMat surface used by the reference:
    `cK \\ y` (GPE.jl:208), logdet (GPE.jl:210), whiten! (GP.jl:27), cholfactors (GP.jl:89)."""

    def __init__(self, ctx, x_colmajor, bits):
        self.ctx = ctx
        self.bits = bits
        d, n = x_colmajor.shape
        self.dim, self.n = d, n
        h = C.c_void_p()
        ctx.check(_lib.load().gpmi_gp_create(ctx.h, bits, d, n, x_colmajor.ctypes.data, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.ctx.h:
                _lib.load().gpmi_gp_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _rhs(self, b):
        dt = _lib.np_dtype(self.bits)
        b = np.array(b, dtype=dt, order="F", copy=True)
        if b.shape[0] != self.n:
            raise _lib.ArgumentError("right-hand side has the wrong number of rows")
        return b

    def solve(self, b):  # cK \ b
        b = self._rhs(b)
        nrhs = 1 if b.ndim == 1 else b.shape[1]
        self.ctx.check(_lib.load().gpmi_solve(self.h, nrhs, b.ctypes.data))
        return b

    def whiten(self, b):  # L⁻¹ b, L = Uᵀ
        b = self._rhs(b)
        nrhs = 1 if b.ndim == 1 else b.shape[1]
        self.ctx.check(_lib.load().gpmi_whiten(self.h, nrhs, b.ctypes.data))
        return b

    def logdet(self):
        out = C.c_double()
        self.ctx.check(_lib.load().gpmi_logdet(self.h, C.byref(out)))
        return out.value

    def inv_diag(self):
        """diag(inv(cK)) — the only part of `inv(Σ)` predict_LOO uses (crossvalidation.jl:8-13)."""
        out = np.empty(self.n, dtype=_lib.np_dtype(self.bits))
        self.ctx.check(_lib.load().gpmi_inv_diag(self.h, out.ctypes.data))
        return out

    def factor_diag(self):
        """diag(cholfactors(cK)) without moving the n × n factor."""
        out = np.empty(self.n, dtype=_lib.np_dtype(self.bits))
        self.ctx.check(_lib.load().gpmi_factor_diag(self.h, out.ctypes.data))
        return out

    def cholfactors(self):
        """Upper factor U (n × n), as Cholesky(factors, 'U', 0) holds it (GPE.jl:60)."""
        U = np.empty((self.n, self.n), dtype=_lib.np_dtype(self.bits), order="F")
        self.ctx.check(_lib.load().gpmi_factor_to_host(self.h, U.ctypes.data))
        return U


class GPE:
    def __init__(self, x, y, mean=None, kernel=None, logNoise=-2.0, covstrat=None, dtype=np.float64, ctx=None):
        """GPE(x, y, mean, kernel, logNoise[, covstrat]) — GPE.jl:68-71.  covstrat None = the exact (dense) device path;
        a FullyIndepStrat selects FITC (gpmi355x.sparse)."""
        self.covstrat = covstrat
        if kernel is None or not isinstance(kernel, Kernel):
            raise _lib.ArgumentError("a Kernel is required")
        self.mean = mean if mean is not None else MeanZero()
        if not isinstance(self.mean, Mean):
            raise _lib.ArgumentError("mean must be a Mean")
        self.kernel = kernel
        self.logNoise = np.asarray(logNoise, dtype=np.float64).copy() if np.ndim(logNoise) else float(logNoise)
        self.bits = 64 if np.dtype(dtype) == np.float64 else 32
        self.ctx = ctx if ctx is not None else _lib.Context.default()
        self.alpha = None
        self.mll = float("nan")
        self.target = float("nan")
        self.fit(x, y)

    # -- fit!(gp, x, y) : GPE.jl:128-138 --------------------------------------
    def fit(self, x, y):
        x = np.asarray(x)
        if x.ndim == 1:
            x = x[None, :]  # x::Vector -> x' row matrix (GPE.jl:96-97)
        y = np.asarray(y, dtype=np.float64)
        if y.ndim != 1 or y.shape[0] != x.shape[1]:
            raise _lib.ArgumentError("Input and output observations must have consistent dimensions.")
        self.x = _lib.colmajor(x, _lib.np_dtype(self.bits))
        self.y = y
        self.dim, self.nobs = self.x.shape
        if self.covstrat is None:
            self.cK = self._alloc_cK()  # alloc_cK (GP.jl:14-20)
        else:
            from .sparse import FullyIndepPDMat, FullyIndepStrat
            if not isinstance(self.covstrat, FullyIndepStrat):
                raise _lib.ArgumentError("covstrat must be None (exact) or a FullyIndepStrat")
            xu = _lib.colmajor(self.covstrat.inducing, _lib.np_dtype(self.bits))
            self.cK = FullyIndepPDMat(self.ctx, self.x, xu, self.bits)  # alloc_cK, fully_indep…:118-132
        self.initialise_target()
        return self

    def _alloc_cK(self):
        """alloc_cK(::CovarianceStrategy, nobs) — src/GP.jl:14-20: the dense device handle (dist.ShardedGPE: a blocked one)"""
        return HIPPDMat(self.ctx, self.x, self.bits)

    # -- update_mll! : GPE.jl:202-212 ------------------------------------------
    def update_mll(self, noise=True, domean=True, kern=True):
        if self.covstrat is not None:
            from .sparse import fitc_update_mll
            return fitc_update_mll(self)
        dt = _lib.np_dtype(self.bits)
        mu = self.mean.mean(self.x)
        ymu = np.ascontiguousarray(self.y - mu, dtype=dt)
        if not (kern or noise) and self.alpha is not None:
            # GPE.jl:203-211: only the mean changed — the factor is kept: alpha = cK \ (y - mu), mll from the stored logdet.  The device copy
            # of alpha (what predict_f and update_dmll read) is replaced by the same call.
            alpha = np.empty(self.nobs, dtype=dt)
            mll = C.c_double()
            self.ctx.check(_lib.load().gpmi_update_alpha(self.cK.h, ymu.ctypes.data, C.byref(mll), alpha.ctypes.data))
            self.alpha = alpha
            self.mll = mll.value
            return self
        ln = np.atleast_1d(np.asarray(self.logNoise, dtype=np.float64))
        if ln.shape[0] not in (1, self.nobs):
            raise _lib.ArgumentError("logNoise must be a scalar or have one entry per observation")
        kd, keep = self.kernel.descriptor(self.dim)
        alpha = np.empty(self.nobs, dtype=dt)
        mll = C.c_double()
        info = C.c_int64()
        rc = _lib.load().gpmi_fit(self.cK.h, C.byref(kd), ln.ctypes.data_as(C.POINTER(C.c_double)), ln.shape[0],
                                  ymu.ctypes.data, C.byref(mll), alpha.ctypes.data, C.byref(info))
        del keep
        self.ctx.check(rc, info.value)
        self.alpha = alpha
        self.mll = mll.value
        return self

    # -- update_dmll! : GPE.jl:298-324 -------------------------------------------
    def update_dmll(self, noise=True, domean=True, kern=True):
        """Gradient of the mll in the order [logNoise; mean…; kernel…] (the exposed parameters only).
        Kernel and noise parts come from the device (gpmi_grad); the mean part is dot(grad_mean, alpha)."""
        if self.alpha is None:
            raise _lib.ArgumentError("update_dmll needs a fitted model (call update_mll first)")
        parts = []
        nfull = self.kernel._full_num_params()
        if noise or kern:
            if np.ndim(self.logNoise) != 0 and noise:
                raise _lib.ArgumentError("the noise gradient needs a scalar logNoise (GPE.jl:313)")
            ln = np.atleast_1d(np.asarray(self.logNoise, dtype=np.float64))
            kd, keep = self.kernel.descriptor(self.dim)
            dk = np.empty(max(nfull, 1), dtype=np.float64)
            dn = C.c_double()
            if self.covstrat is not None:  # FITC: dmll_kern! / dmll_noise of fully_indep_train_conditional.jl:200-257
                rc = _lib.load().gpmi_fitc_grad(self.cK.h, C.byref(kd), float(ln[0]), dk.ctypes.data_as(C.POINTER(C.c_double)), nfull,
                                                C.byref(dn))
            else:
                rc = _lib.load().gpmi_grad(self.cK.h, C.byref(kd), ln.ctypes.data_as(C.POINTER(C.c_double)), ln.shape[0],
                                           dk.ctypes.data_as(C.POINTER(C.c_double)), nfull, C.byref(dn) if noise else None)
            del keep
            self.ctx.check(rc)
        if noise:
            parts.append([dn.value])
        if domean and self.mean.num_params() > 0:
            parts.append(list(self.mean.grad_stack(self.x).T @ np.asarray(self.alpha, dtype=np.float64)))  # GPE.jl:282-288
        if kern:
            parts.append([dk[i] for i in self.kernel.grad_slots()])
        self.dmll = np.asarray([v for p in parts for v in p], dtype=np.float64)
        self.dtarget = self.dmll
        return self

    def update_mll_and_dmll(self, **kw):  # GPE.jl:331-334
        self.update_mll(**kw)
        self.target = self.mll
        return self.update_dmll(**kw)

    # -- target = mll + log prior : GPE.jl:346-392, 514-526 ----------------------
    @property
    def noise_param(self):
        """gp.logNoise as a parameter object: set_priors(gp.noise_param, [Normal(-1.0, 0.5)])."""
        if getattr(self, "_noise_param", None) is None:
            self._noise_param = _priors.NoiseParam(self)
        return self._noise_param

    def _prior_logpdf(self):
        return _priors.prior_logpdf(self.mean) + _priors.prior_logpdf(self.kernel) + _priors.prior_logpdf(self.noise_param)

    def prior_gradlogpdf(self, noise=True, domean=True, kern=True):  # GPE.jl:514-526
        parts = []
        if noise:
            parts.append(_priors.prior_gradlogpdf(self.noise_param))
        if domean:
            parts.append(_priors.prior_gradlogpdf(self.mean))
        if kern:
            parts.append(_priors.prior_gradlogpdf(self.kernel))
        return np.concatenate(parts) if parts else np.zeros(0)

    def update_target_and_dtarget(self, **kw):  # GPE.jl:387-392
        self.update_mll_and_dmll(**kw)
        self.target = self.mll + self._prior_logpdf()
        self.dtarget = self.dmll + self.prior_gradlogpdf(**kw)
        return self

    def initialise_target(self):  # GPE.jl:346-350
        self.update_mll()
        self.target = self.mll + self._prior_logpdf()
        return self

    def update_target(self, **kw):  # GPE.jl:361-365
        self.update_mll(**kw)
        self.target = self.mll + self._prior_logpdf()
        return self

    # -- predict : GP.jl:64-84, GPE.jl:408-416 ---------------------------------
    def predict_f(self, x, full_cov=False):
        x = np.asarray(x)
        if x.ndim == 1:
            x = x[None, :]
        if x.shape[0] != self.dim:
            raise _lib.ArgumentError("Gaussian Process object and input observations do not have consistent dimensions")
        dt = _lib.np_dtype(self.bits)
        xp = _lib.colmajor(x, dt)
        if self.covstrat is not None:
            from .sparse import fitc_predict_f
            return fitc_predict_f(self, xp, full_cov)
        P = xp.shape[1]
        mx = np.ascontiguousarray(self.mean.mean(xp), dtype=dt)
        mu = np.empty(P, dtype=dt)
        var = np.empty((P, P), dtype=dt, order="F") if full_cov else np.empty(P, dtype=dt)
        kd, keep = self.kernel.descriptor(self.dim)
        rc = _lib.load().gpmi_predict(self.cK.h, C.byref(kd), P, xp.ctypes.data, mx.ctypes.data,
                                      1 if full_cov else 0, mu.ctypes.data, var.ctypes.data)
        del keep
        self.ctx.check(rc)
        return mu, var

    # -- leave-one-out : src/crossvalidation.jl:8-13, 31-37, 50-58 ---------------------
    def predict_LOO(self):
        """(μᵢ, σᵢ²) of yᵢ given y₋ᵢ for every observation: σᵢ² = 1 / (Σ⁻¹)ᵢᵢ, μᵢ = yᵢ − αᵢ σᵢ²."""
        if self.covstrat is not None:
            raise _lib.ArgumentError("predict_LOO covers the exact path only")
        s2 = 1.0 / np.asarray(self.cK.inv_diag(), dtype=np.float64)
        return self.y - np.asarray(self.alpha, dtype=np.float64) * s2, s2

    def logp_LOO(self):
        mu, s2 = self.predict_LOO()
        return float(np.sum(-0.5 * np.log(2.0 * np.pi * s2) - 0.5 * (self.y - mu) ** 2 / s2))

    # -- rand(gp, x, n) : src/GP.jl:120-146 (posterior branch) ------------------------
    def rand(self, x, n=1, nugget=1e-10, rng=None):
        """Posterior draws at the columns of x: μ + unwhiten(Σ + nugget·I, randn) with (μ, Σ) = predict_f(full_cov=true).
        The P × P factorisation is host work, like the reference's."""
        rng = rng if rng is not None else np.random.default_rng()
        mu, S = self.predict_f(x, full_cov=True)
        S = np.array(S, dtype=np.float64)
        S[np.diag_indices_from(S)] += nugget
        try:
            L = np.linalg.cholesky(S)
        except np.linalg.LinAlgError as e:
            raise _lib.PosDefException(-1) from e
        out = np.asarray(mu, dtype=np.float64)[:, None] + L @ rng.standard_normal((S.shape[0], n))
        return out

    def noise_variance(self):  # GPE.jl:269-271
        return np.exp(2.0 * np.asarray(self.logNoise))

    def predict_y(self, x, full_cov=False):
        mu, s2 = self.predict_f(x, full_cov=full_cov)
        nv = self.noise_variance()
        if full_cov:
            return mu, s2 + nv * np.eye(s2.shape[0], dtype=s2.dtype)
        return mu, s2 + nv

    # -- parameters : GPE.jl:447-512 -------------------------------------------
    def get_params(self, noise=True, domean=True, kern=True):
        p = []
        if noise:
            p += list(np.atleast_1d(self.logNoise))
        if domean:
            p += list(self.mean.get_params())
        if kern:
            p += list(self.kernel.get_params())
        return [float(v) for v in p]

    def num_params(self, **kw):
        return len(self.get_params(**kw))

    def set_params(self, hyp, noise=True, domean=True, kern=True):
        hyp = [float(v) for v in hyp]
        i = 0
        if noise:
            nn = 1 if np.ndim(self.logNoise) == 0 else len(self.logNoise)
            self.logNoise = hyp[0] if nn == 1 and np.ndim(self.logNoise) == 0 else np.asarray(hyp[:nn])
            i += nn
        nm = self.mean.num_params()
        if domean and nm > 0:
            self.mean.set_params(hyp[i:i + nm])
            i += nm
        if kern:
            nk = self.kernel.num_params()
            self.kernel.set_params(hyp[i:i + nk])
            i += nk


def GP(x, y, mean=None, kernel=None, logNoise=-2.0, packed=False, **kw):
    """GP(x, y, mean, kernel, logNoise) — src/GPE.jl:119-120.

    packed=True: the factor is kept in PACKED storage (stripes of block-rows that stop at their own diagonal — no upper
    triangle, N²/2·(1 + 1/S) elements), which lifts the single-device ceiling from N ≈ 180 000 to
    N ≈ 250 000 in fp64.  comm=<gpmi355x.dist communicator>: the factor row-block sharded over the ranks (one process per GPU).
    Either way the object is a GPE on a BLOCKED handle (gpmi_gp_create_blocked): every GPE verb works, update_dmll / optimize
    included (the blocked gradient needs one more own-rows × N matrix instead of two N × N)."""
    if packed or kw.get("comm") is not None:
        from .dist import ShardedGPE

        return ShardedGPE(x, y, mean, kernel, logNoise, dtype=kw.pop("dtype", np.float64), ctx=kw.pop("ctx", None),
                          block=kw.pop("block", 1024 if packed else None), stripe_blocks=kw.pop("stripe_blocks", 8 if packed else 0), **kw)
    return GPE(x, y, mean, kernel, logNoise, **kw)


def FITC(x, inducing, y, mean=None, kernel=None, logNoise=-2.0, **kw):
    """FITC(x, inducing, y, mean, kernel, logNoise) — src/sparse/fully_indep_train_conditional.jl:333-336."""
    from .sparse import FullyIndepStrat
    return GPE(x, y, mean, kernel, logNoise, covstrat=FullyIndepStrat(inducing), **kw)


def predict_LOO(gp):
    return gp.predict_LOO()


def logp_LOO(gp):
    return gp.logp_LOO()


# functional spellings of the reference's exported verbs
def update_mll(gp, **kw):
    return gp.update_mll(**kw)


def update_target(gp, **kw):
    return gp.update_target(**kw)


def predict_f(gp, x, full_cov=False):
    return gp.predict_f(x, full_cov=full_cov)


def predict_y(gp, x, full_cov=False):
    return gp.predict_y(x, full_cov=full_cov)


def get_params(gp, **kw):
    return gp.get_params(**kw)


def set_params(gp, hyp, **kw):
    return gp.set_params(hyp, **kw)


def optimize_bounds(gp, noisebounds=None, meanbounds=None, kernbounds=None, noise=True, domean=True, kern=True):
    """bounds(gp, ...) — src/GPE.jl:467-490: lower/upper vectors in get_params order (noise, mean, kernel), infinite where
    no pair is given.  Returns None when no bound is given at all (the reference then runs the unconstrained optimizer),
    else a list of (lower, upper) per parameter."""
    if noisebounds is None and meanbounds is None and kernbounds is None:
        return None
    lb, ub = [], []

    def append(n, pair, what):
        if n == 0:
            return
        if pair is None:
            lb.extend([-math.inf] * n)
            ub.extend([math.inf] * n)
            return
        lo, hi = (np.atleast_1d(np.asarray(v, dtype=float)) for v in pair)
        if len(lo) != n or len(hi) != n:
            raise _lib.ArgumentError("%s bounds need %d lower and %d upper values" % (what, n, n))
        lb.extend(lo.tolist())
        ub.extend(hi.tolist())

    if noise:
        append(gp.num_params(noise=True, domean=False, kern=False), noisebounds, "noise")
    if domean:
        append(gp.mean.num_params(), meanbounds, "mean")
    if kern:
        append(gp.kernel.num_params(), kernbounds, "kernel")
    return list(zip(lb, ub))


def optimize(gp, noise=True, domean=True, kern=True, method="L-BFGS-B", options=None, meanbounds=None, kernbounds=None,
             noisebounds=None):
    """optimize!(gp) — src/optimize.jl:19-37 with its error contract (:48-58, :74-83): a PosDefException /
    ArgumentError during an evaluation restores the previous parameters and the point is reported as infeasible
    (Inf, zero gradient).  Target and gradient come from the device (update_target_and_dtarget!, GPE.jl:387-392).
    `noisebounds` / `meanbounds` / `kernbounds` are (lower, upper) pairs of per-parameter vectors as in the reference
    (GPE.jl:467-490: unbounded where a pair is missing); they switch the reference to Fminbox, here they go to the
    bound-constrained L-BFGS-B."""
    from scipy.optimize import minimize

    kw = dict(noise=noise, domean=domean, kern=kern)

    def fg(hyp):
        prev = gp.get_params(**kw)
        try:
            gp.set_params(hyp, **kw)
            gp.update_target_and_dtarget(**kw)
            return -gp.target, -gp.dtarget
        except (_lib.PosDefException, _lib.ArgumentError):
            gp.set_params(prev, **kw)
            return math.inf, np.zeros(len(hyp))

    x0 = np.asarray(gp.get_params(**kw), dtype=float)
    # The starting point is evaluated OUTSIDE the error contract: an ArgumentError that does not depend on the parameter
    # values (a model / size the device gradient does not cover, vector logNoise with noise=True — an @assert in the
    # reference, GPE.jl:304) or a start that is not positive definite must surface, not come back as a "converged"
    # result at x0 with fun = inf.
    gp.update_target_and_dtarget(**kw)
    box = optimize_bounds(gp, noisebounds, meanbounds, kernbounds, **kw)
    res = minimize(fg, x0, jac=True, method=method, bounds=box, options=options or {"maxiter": 20})
    gp.set_params(res.x, **kw)
    gp.update_target()
    return res
