"""Priors on hyperparameters (src/common.jl:118-170, src/GPE.jl:346-392, 514-526): host-side plumbing of
update_target! — target = mll + log prior, dtarget = dmll + d log prior — so that optimize! finds the MAP estimate
exactly as in the reference (test/optim.jl:37-52).  The reference takes Distributions.jl objects; here a prior is any
object with logpdf(x) and gradlogpdf(x): Normal and Uniform below, or a frozen scipy.stats distribution through Scipy()."""
from __future__ import annotations

import math

import numpy as np

from . import _lib


class Normal:
    def __init__(self, mu=0.0, sigma=1.0):
        if not sigma > 0:
            raise _lib.ArgumentError("Normal: sigma must be positive")
        self.mu, self.sigma = float(mu), float(sigma)

    def logpdf(self, x):
        z = (x - self.mu) / self.sigma
        return -0.5 * z * z - math.log(self.sigma) - 0.5 * math.log(2.0 * math.pi)

    def gradlogpdf(self, x):
        return -(x - self.mu) / (self.sigma * self.sigma)


class Uniform:
    def __init__(self, a, b):
        if not b > a:
            raise _lib.ArgumentError("Uniform: needs a < b")
        self.a, self.b = float(a), float(b)

    def logpdf(self, x):
        return -math.log(self.b - self.a) if self.a <= x <= self.b else -math.inf

    def gradlogpdf(self, x):
        return 0.0


class Scipy:
    """Adapter for a frozen scipy.stats distribution (gradient of the log density by central differences)."""

    def __init__(self, frozen, h=1e-6):
        self.d, self.h = frozen, h

    def logpdf(self, x):
        return float(self.d.logpdf(x))

    def gradlogpdf(self, x):
        h = self.h * max(1.0, abs(x))
        return float(self.d.logpdf(x + h) - self.d.logpdf(x - h)) / (2.0 * h)


def _components(obj):
    """Direct components of a composite kernel / mean, else None."""
    from .kernels import _Pair
    from .means import _CompositeMean
    if isinstance(obj, _Pair):
        return [obj.kleft, obj.kright]
    if isinstance(obj, _CompositeMean):
        return list(obj.means)
    return None


def get_priors(obj):
    """common.jl:123-131, pair_kernel.jl:38, masked_kernel.jl:90, fixed_kernel.jl:78-84."""
    from .kernels import FixedKernel, Masked
    comps = _components(obj)
    if comps is not None:
        return [p for c in comps for p in get_priors(c)]
    if isinstance(obj, Masked):
        return get_priors(obj.kernel)
    if isinstance(obj, FixedKernel):
        inner = get_priors(obj.kernel)
        return [inner[i] for i in obj.free] if inner else []
    return list(getattr(obj, "priors", []))


def set_priors(obj, priors):
    """set_priors!(obj, priors) — one prior per exposed parameter, in get_params order (common.jl:133-149)."""
    from .kernels import FixedKernel, Masked
    priors = list(priors)
    if len(priors) != obj.num_params():
        raise _lib.ArgumentError("%s object requires %d priors" % (type(obj).__name__, obj.num_params()))
    comps = _components(obj)
    if comps is not None:
        i = 0
        for c in comps:
            n = c.num_params()
            set_priors(c, priors[i:i + n])
            i += n
    elif isinstance(obj, Masked):
        set_priors(obj.kernel, priors)
    elif isinstance(obj, FixedKernel):  # fixed_kernel.jl:86-90: only the free slots of the wrapped kernel's list change
        n = obj.kernel.num_params()
        inner = get_priors(obj.kernel)
        if len(inner) != n:
            raise _lib.ArgumentError("FixedKernel: set the wrapped kernel's priors before fixing parameters")
        for i, p in zip(obj.free, priors):
            inner[i] = p
        set_priors(obj.kernel, inner)
    else:
        obj.priors = priors


def prior_logpdf(obj):
    """common.jl:151-158; a FixedKernel contributes nothing (fixed_kernel.jl:92-94).  For composites the components'
    contributions are summed, which equals the reference wherever every component carries priors (or none does)."""
    from .kernels import FixedKernel
    if isinstance(obj, FixedKernel) or obj.num_params() == 0:
        return 0.0
    comps = _components(obj)
    if comps is not None:
        return float(sum(prior_logpdf(c) for c in comps))
    pri = get_priors(obj)
    if not pri:
        return 0.0
    return float(sum(p.logpdf(v) for p, v in zip(pri, obj.get_params())))


def prior_gradlogpdf(obj):
    """common.jl:160-167, fixed_kernel.jl:96-98: one entry per exposed parameter."""
    from .kernels import FixedKernel
    n = obj.num_params()
    if n == 0:
        return np.zeros(0)
    if isinstance(obj, FixedKernel):
        return np.zeros(n)
    comps = _components(obj)
    if comps is not None:
        return np.concatenate([prior_gradlogpdf(c) for c in comps])
    pri = get_priors(obj)
    if not pri:
        return np.zeros(n)
    return np.array([p.gradlogpdf(v) for p, v in zip(pri, obj.get_params())], dtype=float)


class NoiseParam:
    """View of gp.logNoise as a parameter object (the reference's Scalar / VectorParam wrapper, common.jl) so that the
    same four functions apply:  set_priors(gp.noise_param, [Normal(-1.0, 0.5)])."""

    def __init__(self, gp):
        # a WEAK reference: the GP owns this view, and a strong reference back would put every GP into a reference cycle —
        # its device buffers (tens of GB) would then live until the cyclic collector happens to run instead of being
        # released when the last reference goes
        import weakref

        self._gp = weakref.ref(gp)
        self.priors = []

    def get_params(self):
        return [float(v) for v in np.atleast_1d(self._gp().logNoise)]

    def num_params(self):
        return len(self.get_params())
