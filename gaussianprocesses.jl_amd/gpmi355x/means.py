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
