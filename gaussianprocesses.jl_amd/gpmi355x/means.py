"""Mean functions (src/means/): evaluated on the HOST — O(N·d), not GPU work
(SURVEY.md §8 a11).  update_mll! only needs μ = mean(m, X) to form y − μ (src/GPE.jl:206-207)."""
from __future__ import annotations

import numpy as np

from . import _lib


class Mean:
    def get_params(self):
        return []

    def set_params(self, hyp):
        if len(hyp) != 0:
            raise _lib.ArgumentError("mean function has no parameters")

    def num_params(self):
        return len(self.get_params())

    def grad_stack(self, X):
        """nobs × num_params matrix of d mean / d parameter (means/means.jl:16-23)."""
        return np.zeros((np.asarray(X).shape[1], 0))


class MeanZero(Mean):  # means/mZero.jl:16
    def mean(self, X):
        return np.zeros(np.asarray(X).shape[1])


class MeanConst(Mean):  # means/mConst.jl:27
    def __init__(self, beta):
        self.beta = float(beta)

    def mean(self, X):
        return np.full(np.asarray(X).shape[1], self.beta)

    def grad_stack(self, X):  # mConst.jl:36
        return np.ones((np.asarray(X).shape[1], 1))

    def get_params(self):
        return [self.beta]

    def set_params(self, hyp):
        if len(hyp) != 1:
            raise _lib.ArgumentError("Constant mean function only has 1 parameter")
        self.beta = float(hyp[0])


class MeanLin(Mean):  # means/mLin.jl:27   X'β
    def __init__(self, beta):
        self.beta = np.asarray(beta, dtype=np.float64).copy()

    def mean(self, X):
        return np.asarray(X, dtype=np.float64).T @ self.beta

    def grad_stack(self, X):  # mLin.jl:38
        return np.asarray(X, dtype=np.float64).T.copy()

    def get_params(self):
        return list(self.beta)

    def set_params(self, hyp):
        if len(hyp) != len(self.beta):
            raise _lib.ArgumentError("Linear mean function: wrong number of parameters")
        self.beta = np.asarray(hyp, dtype=np.float64).copy()


class MeanPoly(Mean):  # means/mPoly.jl:13-33   m(x) = sum_ij beta[i, j] x_i^(j+1), beta is d x degree
    def __init__(self, beta):
        b = np.asarray(beta, dtype=np.float64)
        if b.ndim != 2:
            raise _lib.ArgumentError("MeanPoly needs a d x degree coefficient matrix")
        self.beta = b.copy()

    def _powers(self, X):  # (degree, d, nobs): x_i^j for j = 1..degree
        X = np.asarray(X, dtype=np.float64)
        if X.shape[0] != self.beta.shape[0]:
            raise _lib.ArgumentError("Observations and mean function have inconsistent dimensions")
        return np.stack([X ** (j + 1) for j in range(self.beta.shape[1])])

    def mean(self, X):
        return np.einsum("jin,ij->n", self._powers(X), self.beta)

    def grad_stack(self, X):  # mPoly.jl:48-58: vec of the d x degree matrix x_i^j, column-major like get_params
        P = self._powers(X)
        return P.reshape(P.shape[0] * P.shape[1], P.shape[2]).T.copy()

    def get_params(self):  # vec(beta): column-major
        return list(self.beta.T.reshape(-1))

    def set_params(self, hyp):
        if len(hyp) != self.beta.size:
            raise _lib.ArgumentError("Polynomial mean function has %d parameters" % self.beta.size)
        self.beta = np.asarray(hyp, dtype=np.float64).reshape(self.beta.shape[1], self.beta.shape[0]).T.copy()


class MeanPeriodic(Mean):  # means/mPeriodic.jl:12-38   m(x) = a'cos(2 pi x / p) + b'sin(2 pi x / p), period on the log scale
    def __init__(self, a, b, lp):
        self.a = np.atleast_1d(np.asarray(a, dtype=np.float64)).copy()
        self.b = np.atleast_1d(np.asarray(b, dtype=np.float64)).copy()
        self.p = np.exp(np.atleast_1d(np.asarray(lp, dtype=np.float64)))
        if not (len(self.a) == len(self.b) == len(self.p)):
            raise _lib.ArgumentError("MeanPeriodic: a, b and the periods must have the same length")

    def _phase(self, X):
        X = np.asarray(X, dtype=np.float64)
        if X.shape[0] != len(self.a):
            raise _lib.ArgumentError("Observations and mean function have inconsistent dimensions")
        return 2.0 * np.pi * X / self.p[:, None]

    def mean(self, X):
        ph = self._phase(X)
        return self.a @ np.cos(ph) + self.b @ np.sin(ph)

    def grad_stack(self, X):  # mPeriodic.jl:57-66: [cos; sin; (a sin - b cos) 2 pi x / p]
        ph = self._phase(X)
        c, s = np.cos(ph), np.sin(ph)
        return np.vstack([c, s, (self.a[:, None] * s - self.b[:, None] * c) * ph]).T.copy()

    def get_params(self):
        return list(self.a) + list(self.b) + list(np.log(self.p))

    def set_params(self, hyp):
        d = len(self.a)
        if len(hyp) != 3 * d:
            raise _lib.ArgumentError("MeanPeriodic mean function has %d parameters" % (3 * d))
        h = np.asarray(hyp, dtype=np.float64)
        self.a, self.b, self.p = h[:d].copy(), h[d:2 * d].copy(), np.exp(h[2 * d:])


class _CompositeMean(Mean):  # means/composite_mean.jl: parameters are the components' parameters, concatenated
    def __init__(self, *means):
        if not means or not all(isinstance(m, Mean) for m in means):
            raise _lib.ArgumentError("composite mean needs Mean components")
        self.means = tuple(means)

    def get_params(self):
        return [p for m in self.means for p in m.get_params()]

    def set_params(self, hyp):
        if len(hyp) != self.num_params():
            raise _lib.ArgumentError("composite mean function has %d parameters" % self.num_params())
        i = 0
        for m in self.means:
            n = m.num_params()
            m.set_params(list(hyp[i:i + n]))
            i += n


class SumMean(_CompositeMean):  # means/sum_mean.jl
    def mean(self, X):
        return sum(m.mean(X) for m in self.means)

    def grad_stack(self, X):
        return np.hstack([m.grad_stack(X) for m in self.means])


class ProdMean(_CompositeMean):  # means/prod_mean.jl: d/dtheta of component i is scaled by the product of the other means
    def mean(self, X):
        out = np.ones(np.asarray(X).shape[1])
        for m in self.means:
            out = out * m.mean(X)
        return out

    def grad_stack(self, X):
        vals = [m.mean(X) for m in self.means]
        cols = []
        for i, m in enumerate(self.means):
            others = np.ones(np.asarray(X).shape[1])
            for j, v in enumerate(vals):
                if j != i:
                    others = others * v
            cols.append(m.grad_stack(X) * others[:, None])
        return np.hstack(cols)


def _mean_add(m1, m2):  # sum_mean.jl:24-27: sums flatten
    a = m1.means if isinstance(m1, SumMean) else (m1,)
    b = m2.means if isinstance(m2, SumMean) else (m2,)
    return SumMean(*a, *b)


def _mean_mul(m1, m2):  # prod_mean.jl:31-34: products flatten
    a = m1.means if isinstance(m1, ProdMean) else (m1,)
    b = m2.means if isinstance(m2, ProdMean) else (m2,)
    return ProdMean(*a, *b)


Mean.__add__ = _mean_add
Mean.__mul__ = _mean_mul
