"""CPU oracle for the GaussianProcesses.jl exact-GP fit/predict path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.  The
product path (``gaussianprocesses.jl_amd/``) never imports this module and has
no CPU fallback.

PARITY PINNING.  The reference (Julia) cannot be executed in the build
container, and its test-suite stores no numeric golden for the exact path
(SURVEY.md §8c).  What the reference DOES hold are numbers it printed itself:
the documentation transcripts ``docs/src/Regression.md`` (mll, 2 x 20
predictive means / variances, the optimiser's minimum and minimiser for a
1-d SEIso model; mll and optimum for a 2-d Mat52Ard + SEIso model) and
``docs/src/sparse_example.md`` (exact mll at n = 5000 to 8 digits), plus the
stored FITC value of ``test/test_sparse.jl:156``.  All sit on inputs drawn from
Julia's RNG; ``oracle/julia_mt.py`` restates that RNG and the transcripts print
enough of their inputs to prove the regenerated stream right.  This oracle is
pinned by those reference-produced numbers
(``tests/test_reference_goldens.py``, transcription with file:line in
``tests/golden/reference_transcripts.py``), and additionally by
  (i)  the relational properties the reference tests assert (tests/test_oracle.py
       mirrors test/kernels.jl:39-41,55-60, test/gp.jl:47-53),
  (ii) an independent implementation of the same mathematics (scikit-learn's
       GaussianProcessRegressor, tests/test_oracle.py::test_vs_sklearn_*), and
  (iii) closed forms for N = 1, 2.
Kernels the transcripts do not exercise (Mat12/Mat32, RQ, Noise, Const,
Masked, Fixed, Prod) are pinned by (i)-(iii) only.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Arrays use the reference orientation: ``x`` is ``d × N``
(one observation per COLUMN, src/GPE.jl:41).

Kernel *specs* are nested tuples (0-based ``active_dims``; the reference is
1-based):

    ("se_iso", ll, lsig)            ("se_ard", [ll...], lsig)
    ("mat12_iso", ll, lsig)         ("mat12_ard", [ll...], lsig)
    ("mat32_iso", ll, lsig)         ("mat32_ard", [ll...], lsig)
    ("mat52_iso", ll, lsig)         ("mat52_ard", [ll...], lsig)
    ("rq_iso", ll, lsig, lalpha)    ("rq_ard", [ll...], lsig, lalpha)
    ("noise", lsig)                 ("const", lsig)
    ("sum", k1, k2)                 ("prod", k1, k2)
    ("masked", k, [dims...])        ("fixed", k, [free...])
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

LOG2PI = math.log(2.0 * math.pi)
SQRT3 = math.sqrt(3.0)
SQRT5 = math.sqrt(5.0)
# Julia `x ≈ y` for Float64: rtol = sqrt(eps(Float64)), atol = 0 (Base.isapprox)
ISAPPROX_RTOL = math.sqrt(np.finfo(np.float64).eps)

_ISO = {"se_iso", "mat12_iso", "mat32_iso", "mat52_iso", "rq_iso"}
_ARD = {"se_ard", "mat12_ard", "mat32_ard", "mat52_ard", "rq_ard"}
_SQ = {"se_iso", "se_ard", "rq_iso", "rq_ard"}  # (weighted) squared-euclidean metric


# --------------------------------------------------------------------------
# distances  (src/kernels/distance.jl:41-106, src/kernels/stationary.jl:10-22)
# --------------------------------------------------------------------------
def _sqdist(X1, X2, w=None):
    """r[i,j] = sum_k w_k (X1[k,i]-X2[k,j])^2, accumulated from 0.0 in index
    order k = 1..d exactly as _SqEuclidean_ij / _WeightedSqEuclidean_ij do
    (distance.jl:43-56, 75-88): term = (x-y)^2 * w."""
    d = X1.shape[0]
    r = np.zeros((X1.shape[1], X2.shape[1]), dtype=np.result_type(X1, X2))
    for k in range(d):
        diff = X1[k, :, None] - X2[k, None, :]
        if w is None:
            r += diff * diff
        else:
            r += diff * diff * w[k]
    return r


def _isapprox_cols(X1, X2):
    """Noise kernel δ: all_z X1[z,i] ≈ X2[z,j]  (noise.jl:31-37)."""
    same = np.ones((X1.shape[1], X2.shape[1]), dtype=bool)
    for z in range(X1.shape[0]):
        a = X1[z, :, None]
        b = X2[z, None, :]
        ok = (a == b) | (np.abs(a - b) <= ISAPPROX_RTOL * np.maximum(np.abs(a), np.abs(b)))
        same &= ok
    return same


# --------------------------------------------------------------------------
# leaf kernels:  cov(k, r)   (SURVEY §8 a7 — one line per leaf file)
# --------------------------------------------------------------------------
def _leaf(spec, X1, X2):
    name = spec[0]
    dt = np.result_type(X1, X2)
    if name == "noise":  # noise.jl:27-39
        s2 = math.exp(2.0 * spec[1])
        return np.where(_isapprox_cols(X1, X2), dt.type(s2), dt.type(0.0))
    if name == "const":  # const.jl:25,36
        s2 = math.exp(2.0 * spec[1])
        return np.full((X1.shape[1], X2.shape[1]), s2, dtype=dt)
    if name in _ISO:
        ll, ls = float(spec[1]), float(spec[2])
        s2 = math.exp(2.0 * ls)
        r = _sqdist(X1, X2)
        if name == "se_iso":  # se_iso.jl:28,39   σ2*exp(-0.5*r/ℓ2)
            l2 = math.exp(2.0 * ll)
            return s2 * np.exp(-0.5 * r / l2)
        if name == "rq_iso":  # rq_iso.jl:33,44  σ2*(1+r/(2αℓ2))^(-α)
            l2 = math.exp(2.0 * ll)
            al = math.exp(float(spec[3]))
            return s2 * (1.0 + r / (2.0 * al * l2)) ** (-al)
        r = np.sqrt(r)  # Euclidean metric (distance.jl:64-71)
        ell = math.exp(ll)
        if name == "mat12_iso":  # mat12_iso.jl:30,41
            return s2 * np.exp(-r / ell)
        if name == "mat32_iso":  # mat32_iso.jl:30,41-42
            s = SQRT3 * r / ell
            return s2 * (1.0 + s) * np.exp(-s)
        if name == "mat52_iso":  # mat52_iso.jl:30,40-41
            s = SQRT5 * r / ell
            return s2 * (1.0 + s + s * s / 3.0) * np.exp(-s)
    if name in _ARD:
        ll = np.asarray(spec[1], dtype=np.float64)
        if ll.shape[0] != X1.shape[0]:
            raise ValueError("ARD kernel dimension mismatch")
        s2 = math.exp(2.0 * float(spec[2]))
        w = np.exp(-2.0 * ll)  # iℓ2, se_ard.jl:31
        r = _sqdist(X1, X2, w)
        if name == "se_ard":  # se_ard.jl:43   σ2*exp(-r/2)
            return s2 * np.exp(-r / 2.0)
        if name == "rq_ard":  # rq_ard.jl:47   σ2*(1+0.5*r/α)^(-α)
            al = math.exp(float(spec[3]))
            return s2 * (1.0 + 0.5 * r / al) ** (-al)
        r = np.sqrt(r)  # WeightedEuclidean (distance.jl:99-106)
        if name == "mat12_ard":  # mat12_ard.jl:43
            return s2 * np.exp(-r)
        if name == "mat32_ard":  # mat32_ard.jl:43-44
            s = SQRT3 * r
            return s2 * (1.0 + s) * np.exp(-s)
        if name == "mat52_ard":  # mat52_ard.jl:43-44
            s = SQRT5 * r
            return s2 * (1.0 + s + s * s / 3.0) * np.exp(-s)
    raise ValueError(f"unknown kernel spec {name!r}")


def cov(spec, X1, X2=None):
    """cov(k, X1, X2) / cov(k, X)  — src/kernels/kernels.jl:31-71.

    Returns the nobs1 × nobs2 matrix cK[i,j] = cov_ij(k, X1, X2, i, j).
    """
    X1 = np.asarray(X1)
    X2 = X1 if X2 is None else np.asarray(X2)
    if X1.shape[0] != X2.shape[0]:
        raise ValueError("X1 and X2 must have same dimension")  # kernels.jl:34
    name = spec[0]
    if name == "sum":  # sum_kernel.jl:15-16
        return cov(spec[1], X1, X2) + cov(spec[2], X1, X2)
    if name == "prod":  # prod_kernel.jl:14-15
        return cov(spec[1], X1, X2) * cov(spec[2], X1, X2)
    if name == "masked":  # masked_kernel.jl:44-49 (view of the active rows)
        dims = list(spec[2])
        return cov(spec[1], X1[dims, :], X2[dims, :])
    if name == "fixed":  # fixed_kernel.jl:69
        return cov(spec[1], X1, X2)
    return _leaf(spec, X1, X2)


def cov_scalar(spec, x, y):
    """cov(k, x::Vector, y::Vector) — the pairwise definition used by
    test/kernels.jl:41,57 as the meaning of cov!; pure-Python scalar loops."""
    name = spec[0]
    if name == "sum":
        return cov_scalar(spec[1], x, y) + cov_scalar(spec[2], x, y)
    if name == "prod":
        return cov_scalar(spec[1], x, y) * cov_scalar(spec[2], x, y)
    if name == "masked":
        dims = list(spec[2])
        return cov_scalar(spec[1], [x[i] for i in dims], [y[i] for i in dims])
    if name == "fixed":
        return cov_scalar(spec[1], x, y)
    d = len(x)
    if name == "noise":
        s2 = math.exp(2.0 * spec[1])
        for a, b in zip(x, y):
            if not (a == b or abs(a - b) <= ISAPPROX_RTOL * max(abs(a), abs(b))):
                return 0.0
        return s2
    if name == "const":
        return math.exp(2.0 * spec[1])
    if name in _ISO:
        w = [1.0] * d
    else:
        w = [math.exp(-2.0 * float(v)) for v in spec[1]]
    r = 0.0
    for k in range(d):
        r += (x[k] - y[k]) ** 2 * w[k]
    s2 = math.exp(2.0 * float(spec[2]))
    if name == "se_iso":
        return s2 * math.exp(-0.5 * r / math.exp(2.0 * spec[1]))
    if name == "se_ard":
        return s2 * math.exp(-r / 2.0)
    if name == "rq_iso":
        al = math.exp(spec[3])
        return s2 * (1.0 + r / (2.0 * al * math.exp(2.0 * spec[1]))) ** (-al)
    if name == "rq_ard":
        al = math.exp(spec[3])
        return s2 * (1.0 + 0.5 * r / al) ** (-al)
    r = math.sqrt(r)
    if name.endswith("_iso"):
        r = r / math.exp(spec[1])
    if name.startswith("mat12"):
        return s2 * math.exp(-r)
    if name.startswith("mat32"):
        s = SQRT3 * r
        return s2 * (1.0 + s) * math.exp(-s)
    if name.startswith("mat52"):
        s = SQRT5 * r
        return s2 * (1.0 + s + s * s / 3.0) * math.exp(-s)
    raise ValueError(name)


def num_params(spec):
    """num_params(k) — leaf files + pair_kernel.jl:14."""
    name = spec[0]
    if name in ("sum", "prod"):
        return num_params(spec[1]) + num_params(spec[2])
    if name == "masked":
        return num_params(spec[1])
    if name == "fixed":
        return len(spec[2])
    if name in ("noise", "const"):
        return 1
    if name in _ISO:
        return 3 if name == "rq_iso" else 2
    n = len(spec[1]) + 1
    return n + 1 if name == "rq_ard" else n


# --------------------------------------------------------------------------
# means  (src/means/means.jl:6-13, mZero.jl:16, mConst.jl:27, mLin.jl:27)
# --------------------------------------------------------------------------
def mean(mspec, X):
    X = np.asarray(X)
    name = mspec[0]
    if name == "zero":
        return np.zeros(X.shape[1])
    if name == "const":
        return np.full(X.shape[1], float(mspec[1]))
    if name == "lin":
        return X.T @ np.asarray(mspec[1], dtype=np.float64)
    raise ValueError(name)


# --------------------------------------------------------------------------
# fit:  update_cK! + update_mll!   (src/GPE.jl:169-212, src/GP.jl:101-112)
# --------------------------------------------------------------------------
class NotPosDef(Exception):
    """LinearAlgebra.PosDefException(info) analogue (GP.jl:110)."""

    def __init__(self, info):
        super().__init__(f"matrix is not positive definite; Cholesky factorization failed (info={info})")
        self.info = info


def update_cK(spec, x, log_noise):
    """update_cK! (GPE.jl:169-186) + make_posdef! (GP.jl:101-112).

    Returns (Σ, U) with Σ = cov!(x,x) + nugget on the diagonal and
    Σ = UᵀU (upper factor, dpotrf 'U' as cholesky!(Symmetric(·,:U)))."""
    x = np.asarray(x, dtype=np.float64)
    K = cov(spec, x)
    n = K.shape[0]
    if np.ndim(log_noise) == 0:
        K[np.diag_indices(n)] += math.exp(2.0 * float(log_noise))  # GPE.jl:173, GP.jl:104-108
    else:
        ln = np.asarray(log_noise, dtype=np.float64)
        if ln.shape != (n,):
            raise ValueError("heteroscedastic logNoise must have length nobs")
        K[np.diag_indices(n)] += np.exp(2.0 * ln)  # GPE.jl:181-183
    # dpotrf('U') — the routine Julia's cholesky! dispatches to (GP.jl:110)
    U, info = sla.lapack.dpotrf(K, lower=0, clean=1, overwrite_a=0)
    if info != 0:
        raise NotPosDef(int(info))
    return K, U


def update_mll(spec, x, y, log_noise, mspec=("zero",)):
    """update_mll! (GPE.jl:202-212).  Returns dict(mll, alpha, U, K)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if y.shape[0] != x.shape[1]:
        raise ValueError("Input and output observations must have consistent dimensions.")  # GPE.jl:42
    K, U = update_cK(spec, x, log_noise)
    ym = y - mean(mspec, x)  # GPE.jl:206-207
    alpha = sla.cho_solve((U, False), ym)  # GPE.jl:208  (PDMats `\` = dpotrs)
    logdet = 2.0 * np.sum(np.log(np.diag(U)))  # PDMats.logdet
    mll = -(float(ym @ alpha) + logdet + LOG2PI * x.shape[1]) / 2.0  # GPE.jl:210
    return {"mll": mll, "alpha": alpha, "U": U, "K": K, "logdet": logdet}


# --------------------------------------------------------------------------
# predict  (src/GP.jl:25-84, src/GPE.jl:399-416)
# --------------------------------------------------------------------------
def predict_full(spec, x, fit, xpred, mspec=("zero",)):
    """predictMVN (GP.jl:39-49) + predictMVN! (GP.jl:25-30)."""
    x = np.asarray(x, dtype=np.float64)
    xpred = np.asarray(xpred, dtype=np.float64)
    Kcross = cov(spec, x, xpred)  # GP.jl:44
    Kpred = cov(spec, xpred)  # GP.jl:45  (X1 === X2 branch)
    mx = mean(mspec, xpred)
    mu = mx + Kcross.T @ fit["alpha"]  # GP.jl:26
    # whiten!(Kff, Kfx) = L⁻¹ Kfx with L = Uᵀ  (GP.jl:27)
    Lck = sla.solve_triangular(fit["U"], Kcross, trans="T", lower=False)
    Sigma = Kpred - Lck.T @ Lck  # GP.jl:51-54 (syrk + copytri ⇒ symmetric)
    Sigma = np.triu(Sigma) + np.triu(Sigma, 1).T
    return mu, Sigma


def predict_f(spec, x, fit, xpred, mspec=("zero",), full_cov=False, pointwise=False):
    """predict_f (GP.jl:64-79).

    full_cov=False is, in the reference, a loop of P single-point
    predict_full calls with σ² clamped at 0 (GP.jl:69-77).  ``pointwise=True``
    runs literally that loop; the default computes the same numbers batched
    (one triangular solve with P right-hand sides)."""
    x = np.asarray(x, dtype=np.float64)
    xpred = np.asarray(xpred, dtype=np.float64)
    if xpred.shape[0] != x.shape[0]:
        raise ValueError("Gaussian Process object and input observations do not have consistent dimensions")
    if full_cov:
        return predict_full(spec, x, fit, xpred, mspec)
    P = xpred.shape[1]
    if pointwise:
        mu = np.empty(P)
        s2 = np.empty(P)
        for k in range(P):
            m, sig = predict_full(spec, x, fit, xpred[:, k : k + 1], mspec)
            mu[k] = m[0]
            s2[k] = max(sig[0, 0], 0.0)
        return mu, s2
    Kcross = cov(spec, x, xpred)
    mu = mean(mspec, xpred) + Kcross.T @ fit["alpha"]
    Lck = sla.solve_triangular(fit["U"], Kcross, trans="T", lower=False)
    # prior variance = cov(k, xp, xp)[1,1] of each single-column slice (GP.jl:45,73)
    kdiag = _kdiag(spec, xpred)
    s2 = np.maximum(kdiag - np.sum(Lck * Lck, axis=0), 0.0)
    return mu, s2


def _kdiag(spec, X):
    """diag(cov(k, X)) without forming the P×P matrix."""
    name = spec[0]
    P = X.shape[1]
    if name == "sum":
        return _kdiag(spec[1], X) + _kdiag(spec[2], X)
    if name == "prod":
        return _kdiag(spec[1], X) * _kdiag(spec[2], X)
    if name == "masked":
        return _kdiag(spec[1], X[list(spec[2]), :])
    if name == "fixed":
        return _kdiag(spec[1], X)
    if name in ("noise", "const"):
        return np.full(P, math.exp(2.0 * spec[1]))
    return np.full(P, math.exp(2.0 * float(spec[2])))  # every stationary leaf: k(0) = σ2


def predict_y(spec, x, fit, xpred, log_noise, mspec=("zero",), full_cov=False):
    """predict_y (GPE.jl:408-416): predict_f + noise_variance."""
    mu, s2 = predict_f(spec, x, fit, xpred, mspec, full_cov=full_cov)
    nv = math.exp(2.0 * float(log_noise))
    if full_cov:
        return mu, s2 + nv * np.eye(s2.shape[0])
    return mu, s2 + nv


# --------------------------------------------------------------------------
# synthetic workload of SURVEY.md §8(d)  (shared by tests and bench.py's
# cpu_baseline leg so CPU and GPU see byte-identical inputs)
# --------------------------------------------------------------------------
def synthetic_inputs(n, d, p=1024, seed=20240501):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 1.0, size=(d, n))
    f = np.sin(2.0 * np.pi * x).sum(axis=0) / d
    y = f + 0.1 * rng.standard_normal(n)
    xpred = rng.uniform(0.0, 1.0, size=(d, p))
    return x, y, xpred


# --------------------------------------------------------------------------
# gradient path:  update_dmll!  (src/GPE.jl:151-164, 219-241, 273-324) — SURVEY §8(f-1)
# --------------------------------------------------------------------------
def grad_cov(spec, X):
    """K = cov(k, X) and the stack [dK/dθ_p] in get_params order (log-scale parameters).

    Leaf derivatives follow dk_dθp / dKij_dθp of each leaf file:
      se_iso.jl:41-50  se_ard.jl:45-54  mat12_iso.jl:42 mat12_ard.jl:44  mat32_iso.jl:43-44 mat32_ard.jl:45
      mat52_iso.jl:42-43 mat52_ard.jl:45-46  mat.jl:5-33 (zero at r = 0 / wdiff = 0)  rq_iso.jl:45-61 rq_ard.jl:48-63
      noise.jl:47-48  const.jl:40-46;  Sum: sum_kernel.jl:18-51;  Prod: prod_kernel.jl:17-68;
      Masked: masked_kernel.jl:51-56;  Fixed: fixed_kernel.jl:63-66;  dk_dlσ = 2k: stationary.jl:28."""
    X = np.asarray(X, dtype=np.float64)
    name = spec[0]
    n = X.shape[1]
    if name == "sum":
        K1, d1 = grad_cov(spec[1], X)
        K2, d2 = grad_cov(spec[2], X)
        return K1 + K2, d1 + d2
    if name == "prod":
        K1, d1 = grad_cov(spec[1], X)
        K2, d2 = grad_cov(spec[2], X)
        return K1 * K2, [d * K2 for d in d1] + [K1 * d for d in d2]
    if name == "masked":
        return grad_cov(spec[1], X[list(spec[2]), :])
    if name == "fixed":
        K, d = grad_cov(spec[1], X)
        return K, [d[i] for i in spec[2]]
    if name == "noise":
        K = _leaf(spec, X, X)
        return K, [2.0 * K]
    if name == "const":
        K = _leaf(spec, X, X)
        return K, [2.0 * K]
    K = _leaf(spec, X, X)
    s2 = math.exp(2.0 * float(spec[2]))
    if name in _ISO:
        ll = float(spec[1])
        r = _sqdist(X, X)
        if name == "se_iso":
            return K, [r / math.exp(2.0 * ll) * K, 2.0 * K]
        if name == "rq_iso":
            al = math.exp(float(spec[3]))
            s = r / math.exp(2.0 * ll)
            part = 1.0 + s / (2.0 * al)
            return K, [s2 * s * part ** (-al - 1.0), 2.0 * K, s2 * part ** (-al) * (s / (2.0 * part) - al * np.log(part))]
        r = np.sqrt(r)
        ell = math.exp(ll)
        if name == "mat12_iso":
            dll = r / ell * K
        elif name == "mat32_iso":
            s = SQRT3 * r / ell
            dll = s2 * s * s * np.exp(-s)
        else:
            s = SQRT5 * r / ell
            dll = s2 / 3.0 * s * s * (1.0 + s) * np.exp(-s)
        return K, [np.where(r == 0.0, 0.0, dll), 2.0 * K]
    lls = np.asarray(spec[1], dtype=np.float64)
    w = np.exp(-2.0 * lls)
    r = _sqdist(X, X, w)
    wd = [(X[k, :, None] - X[k, None, :]) ** 2 * w[k] for k in range(X.shape[0])]
    if name == "se_ard":
        return K, [wdk * K for wdk in wd] + [2.0 * K]
    if name == "rq_ard":
        al = math.exp(float(spec[3]))
        part = 1.0 + r / (2.0 * al)
        return K, [s2 * wdk * part ** (-al - 1.0) for wdk in wd] + [2.0 * K, s2 * part ** (-al) * (r / (2.0 * part) - al * np.log(part))]
    re = np.sqrt(r)
    with np.errstate(divide="ignore", invalid="ignore"):
        if name == "mat12_ard":
            dl = [np.where(wdk > 0, wdk / re * K, 0.0) for wdk in wd]
        elif name == "mat32_ard":
            dl = [np.where(wdk > 0, 3.0 * s2 * wdk * np.exp(-SQRT3 * re), 0.0) for wdk in wd]
        else:
            s = SQRT5 * re
            dl = [np.where(wdk > 0, 5.0 / 3.0 * s2 * wdk * (1.0 + s) * np.exp(-s), 0.0) for wdk in wd]
    return K, dl + [2.0 * K]


def grad_mean(mspec, X):
    """grad_stack(m, X): nobs × num_params (means/means.jl:16-23; mConst.jl:36, mLin.jl:38)."""
    X = np.asarray(X, dtype=np.float64)
    if mspec[0] == "zero":
        return np.zeros((X.shape[1], 0))
    if mspec[0] == "const":
        return np.ones((X.shape[1], 1))
    if mspec[0] == "lin":
        return X.T.copy()
    raise ValueError(mspec[0])


def update_dmll(spec, x, y, log_noise, mspec=("zero",), fit=None):
    """update_dmll! (GPE.jl:298-324): gradient of the mll in the order [logNoise; mean…; kernel…]."""
    x = np.asarray(x, dtype=np.float64)
    if fit is None:
        fit = update_mll(spec, x, y, log_noise, mspec)
    n = x.shape[1]
    alpha = fit["alpha"]
    Kinv = sla.cho_solve((fit["U"], False), np.eye(n))
    W = np.outer(alpha, alpha) - Kinv  # get_ααinvcKI!, GPE.jl:151-164
    dnoise = math.exp(2.0 * float(log_noise)) * np.trace(W)  # GPE.jl:273-275
    dmean = grad_mean(mspec, x).T @ alpha  # GPE.jl:282-288
    _, dKs = grad_cov(spec, x)
    dkern = np.array([0.5 * np.sum(W * dK) for dK in dKs])  # GPE.jl:219-241 (diag/2 + strict lower = half the full sum)
    return {"dmll": np.concatenate([[dnoise], dmean, dkern]), "dnoise": dnoise, "dmean": dmean, "dkern": dkern}


# --------------------------------------------------------------------------
# FITC: src/sparse/fully_indep_train_conditional.jl, with the pieces it inherits from
# subsetofregressors.jl (alpha_u, predictMVN) and determ_train_conditional.jl (the DTC predictive covariance)
# --------------------------------------------------------------------------
def _chol_upper(m, nugget):
    """make_posdef!(m, factors; nugget) (GP.jl:101-112): add the nugget to the diagonal, upper Cholesky."""
    a = np.array(m, dtype=np.float64)
    if nugget > 0:
        a[np.diag_indices_from(a)] += nugget
    try:
        return a, sla.cholesky(a, lower=False)
    except sla.LinAlgError as e:  # LAPACK info -> PosDefException
        raise NotPosDef(str(e)) from e


def fitc_update_cK(spec, x, xu, log_noise):
    """update_cK!(::FullyIndepPDMat, …) — fully_indep_train_conditional.jl:134-156."""
    x = np.asarray(x, dtype=np.float64)
    xu = np.asarray(xu, dtype=np.float64)
    Kuu, Uuu = _chol_upper(cov(spec, xu), 1e-10)                      # :139-141
    Kuf = cov(spec, xu, x)                                            # :142
    Kdiag = _kdiag(spec, x)                                           # :146
    Luf = sla.solve_triangular(Uuu, Kuf, trans="T", lower=False)      # invquad(Kuu, Kuf[:, i]) = |Uuu^-T Kuf[:, i]|^2
    Qdiag = np.sum(Luf * Luf, axis=0)                                 # :147
    lam = math.exp(2.0 * float(log_noise)) + Kdiag - Qdiag            # :148
    SQR = Kuf @ (Kuf / lam).T + Kuu                                   # :150
    SQR = np.triu(SQR) + np.triu(SQR, 1).T                            # copytri!(…, 'U') :151
    SQR, Usqr = _chol_upper(SQR, 1e-10)                               # :153
    return {"Kuu": Kuu, "Uuu": Uuu, "Kuf": Kuf, "lam": lam, "SQR": SQR, "Usqr": Usqr}


def fitc_solve(cK, b):
    r"""`\`(a::FullyIndepPDMat, x) — fully_indep_train_conditional.jl:38-41."""
    Lk = sla.solve_triangular(cK["Usqr"], cK["Kuf"], trans="T", lower=False)  # whiten(ΣQR, Kuf)
    bl = (b.T / cK["lam"]).T
    return ((b - Lk.T @ (Lk @ bl)).T / cK["lam"]).T


def fitc_logdet(cK):
    """logdet(::FullyIndepPDMat) — fully_indep_train_conditional.jl:80."""
    ld = lambda U: 2.0 * float(np.sum(np.log(np.diag(U))))
    return ld(cK["Usqr"]) - ld(cK["Uuu"]) + float(np.sum(np.log(cK["lam"])))


def fitc_dense(cK):
    """Base.Matrix(::FullyIndepPDMat) — fully_indep_train_conditional.jl:81-86 (what test_sparse.jl:117-132 compares with)."""
    Lk = sla.solve_triangular(cK["Uuu"], cK["Kuf"], trans="T", lower=False)
    return Lk.T @ Lk + np.diag(cK["lam"])


def fitc_update_mll(spec, x, xu, y, log_noise, mspec=("zero",)):
    """update_mll! (GPE.jl:202-212) on a FITC covariance.  Returns the cK pieces + alpha, alpha_u, mll, logdet."""
    x = np.asarray(x, dtype=np.float64)
    cK = fitc_update_cK(spec, x, xu, log_noise)
    ym = np.asarray(y, dtype=np.float64) - mean(mspec, x)
    alpha = fitc_solve(cK, ym)
    logdet = fitc_logdet(cK)
    mll = -(float(ym @ alpha) + logdet + LOG2PI * x.shape[1]) / 2.0
    # get_alpha_u (fully_indep_train_conditional.jl:279-286): ΣQR \ (Kuf (Λ \ (y - m)))
    alpha_u = sla.cho_solve((cK["Usqr"], False), cK["Kuf"] @ (ym / cK["lam"]))
    out = dict(cK)
    out.update({"alpha": alpha, "alpha_u": alpha_u, "mll": mll, "logdet": logdet})
    return out


def fitc_predict_full(spec, xu, fit, xpred, mspec=("zero",)):
    """predictMVN(…, ::FullyIndepStrat, …) = the DTC formulas (fully_indep…:321-329 -> determ_train_conditional.jl:41-59
    -> subsetofregressors.jl:303-321): mu = m(x*) + Kux' alpha_u ;  Sigma = Kxx - Qxx + Kxu ΣQR^-1 Kux."""
    xpred = np.asarray(xpred, dtype=np.float64)
    Kux = cov(spec, np.asarray(xu, dtype=np.float64), xpred)
    mu = mean(mspec, xpred) + Kux.T @ fit["alpha_u"]
    Lck = sla.solve_triangular(fit["Usqr"], Kux, trans="T", lower=False)
    S_sor = Lck.T @ Lck
    Lq = sla.solve_triangular(fit["Uuu"], Kux, trans="T", lower=False)
    Qxx = Lq.T @ Lq                                                    # Xt_invA_X(Kuu, Kux)
    S = cov(spec, xpred) - Qxx + S_sor
    S = np.triu(S) + np.triu(S, 1).T
    return mu, S


def fitc_predict_f(spec, xu, fit, xpred, mspec=("zero",), full_cov=False):
    """predict_f (GP.jl:64-79) on the FITC model: per point max(diag, 0) unless full_cov."""
    mu, S = fitc_predict_full(spec, xu, fit, xpred, mspec)
    if full_cov:
        return mu, S
    return mu, np.maximum(np.diag(S), 0.0)


def grad_cov_rect(spec, Xa, Xb):
    """(cov(k, Xa, Xb), [d cov / dθ_p]) for two point sets, and the diagonal derivative d k(x, x) / dθ_p of Xb's points:
    grad_slice! (kernels.jl:140-160) over a rectangular block and dKij_dθp(kernel, X, X, EmptyData(), i, i, p, dim).
    Evaluated as blocks of grad_cov on the joined set (small cases only)."""
    Xa = np.asarray(Xa, dtype=np.float64)
    Xb = np.asarray(Xb, dtype=np.float64)
    ma = Xa.shape[1]
    K, dK = grad_cov(spec, np.concatenate([Xa, Xb], axis=1))
    return K[:ma, ma:], [d[:ma, ma:] for d in dK], [np.diag(d)[ma:].copy() for d in dK]


def fitc_update_dmll(spec, x, xu, y, log_noise, mspec=("zero",), fit=None):
    """update_dmll! (GPE.jl:298-324) on a FITC model: [dmll_noise; dmll_mean; dmll_kern] as the reference forms them.
      precompute!          subsetofregressors.jl:141-151   Kuu⁻¹Kuf, Kuu⁻¹KufΣ⁻¹y, Σ⁻¹Kfu
      dmll_kern! (SoR)     subsetofregressors.jl:219-256   V = 2 α'∂Kfu b − b'∂Kuu b ;  T = 2 tr(Kuu⁻¹KufΣ⁻¹∂Kfu) − tr(Σ⁻¹Kfu Kuu⁻¹∂Kuu Kuu⁻¹Kuf)
      dmll_kern! (FITC)    fully_indep_train_conditional.jl:200-234   ∂Λ_i = ∂K_ii + a_i'∂Kuu a_i − 2 ∂K_ui'a_i ;  += (α'∂Λα − tr(Σ⁻¹∂Λ)) / 2
      trinvAB              fully_indep_train_conditional.jl:63-67
      dmll_noise           fully_indep_train_conditional.jl:243-257   σ² (α'α − tr Λ⁻¹ + |Lk|²),  Lk = ΣQR^-½ Kuf Λ⁻¹
      dmll_mean!           GPE.jl:282-288"""
    x = np.asarray(x, dtype=np.float64)
    xu = np.asarray(xu, dtype=np.float64)
    if fit is None:
        fit = fitc_update_mll(spec, x, xu, y, log_noise, mspec)
    alpha, lam, Kuf = fit["alpha"], fit["lam"], fit["Kuf"]
    n, m = x.shape[1], xu.shape[1]
    # precompute!
    A = sla.cho_solve((fit["Uuu"], False), Kuf)                       # Kuu⁻¹Kuf                  (m × n)
    b = sla.cho_solve((fit["Uuu"], False), Kuf @ alpha)               # Kuu⁻¹KufΣ⁻¹y              (m)
    SiKfu = fitc_solve(fit, Kuf.T.copy())                             # Σ⁻¹Kfu = cK \ Kfu         (n × m)
    # trinvAB's L
    L = sla.solve_triangular(fit["Usqr"], Kuf / lam, trans="T", lower=False)   # whiten(ΣQR, Kuf Λ⁻¹)
    _, dKuf, dKdiag = grad_cov_rect(spec, xu, x)
    _, dKuu = grad_cov(spec, xu)
    dk = np.zeros(len(dKuf))
    for p in range(len(dKuf)):
        dKuf_p, dKuu_p = dKuf[p], dKuu[p]
        V = 2.0 * float(alpha @ (dKuf_p.T @ b)) - float(b @ (dKuu_p @ b))
        T = 2.0 * float(np.sum(fitc_solve(fit, dKuf_p.T.copy()) * A.T))
        T -= float(np.sum(SiKfu.T * (sla.cho_solve((fit["Uuu"], False), dKuu_p) @ A)))
        dk[p] = (V - T) / 2.0
        dlam = dKdiag[p] + np.einsum("ui,uv,vi->i", A, dKuu_p, A) - 2.0 * np.sum(dKuf_p * A, axis=0)
        V2 = float(alpha @ (dlam * alpha))
        T2 = float(np.sum(dlam / lam)) - float(np.sum(L * (L * dlam)))
        dk[p] += (V2 - T2) / 2.0
    Lk = sla.solve_triangular(fit["Usqr"], Kuf, trans="T", lower=False) / lam
    dnoise = math.exp(2.0 * float(log_noise)) * (float(alpha @ alpha) - float(np.sum(1.0 / lam)) + float(np.sum(Lk * Lk)))
    dmean = grad_mean(mspec, x).T @ alpha                              # dmll_mean!: grad_stack' * alpha
    return {"dmll": np.concatenate([[dnoise], dmean, dk]), "dnoise": dnoise, "dmean": dmean, "dkern": dk}


def _chol_lower_ld(A):
    """Unblocked lower Cholesky in np.longdouble (80-bit on x86): the arithmetic-independent value of a factorisation."""
    LD = np.longdouble
    A = np.asarray(A, dtype=LD)
    n = A.shape[0]
    L = np.zeros_like(A)
    for j in range(n):
        d = A[j, j] - np.dot(L[j, :j], L[j, :j])
        if not d > 0:
            raise NotPosDef(f"pivot {j + 1}")
        L[j, j] = np.sqrt(d)
        if j + 1 < n:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
    return L


def fitc_update_mll_extended(spec, x, xu, y, log_noise, mspec=("zero",)):
    """The same statements as fitc_update_cK / fitc_update_mll (fully_indep_train_conditional.jl:134-156, :38-41, :80,
    :279-286) evaluated in 80-bit arithmetic on the fp64 covariance entries.  ΣQR = Kuf Λ⁻¹ Kfu + Kuu has a condition number
    of order n / (σ² · 1e-10): its small pivots — and with them logdet and alpha_u — are rounding noise in ANY fp64 Cholesky,
    LAPACK's included, so for covariances that are numerically rank-deficient (smooth kernels, many inducing points) this is
    the value a faithful implementation has to be compared with.  O(n m²) Python-level loops: small cases only."""
    LD = np.longdouble
    x = np.asarray(x, dtype=np.float64)
    xu = np.asarray(xu, dtype=np.float64)
    m = xu.shape[1]
    Kuu = cov(spec, xu).astype(LD)
    Kuu[np.diag_indices_from(Kuu)] += LD(1e-10)
    Luu = _chol_lower_ld(Kuu)
    Kuf = cov(spec, xu, x).astype(LD)
    W = np.zeros_like(Kuf)
    for i in range(m):
        W[i] = (Kuf[i] - Luu[i, :i] @ W[:i]) / Luu[i, i]
    lam = LD(math.exp(2.0 * float(log_noise))) + _kdiag(spec, x).astype(LD) - (W * W).sum(axis=0)
    S = Kuf @ (Kuf / lam).T + Kuu
    S[np.diag_indices_from(S)] += LD(1e-10)
    LS = _chol_lower_ld(S)
    r = (np.asarray(y, dtype=np.float64) - mean(mspec, x)).astype(LD)
    t = Kuf @ (r / lam)
    z = np.zeros(m, dtype=LD)
    for i in range(m):
        z[i] = (t[i] - LS[i, :i] @ z[:i]) / LS[i, i]
    au = np.zeros(m, dtype=LD)
    for i in range(m - 1, -1, -1):
        au[i] = (z[i] - LS[i + 1:, i] @ au[i + 1:]) / LS[i, i]
    alpha = (r - Kuf.T @ au) / lam
    logdet = 2 * np.log(np.diag(LS)).sum() - 2 * np.log(np.diag(Luu)).sum() + np.log(lam).sum()
    mll = -(r @ alpha + logdet + LD(LOG2PI) * x.shape[1]) / 2
    return {"mll": float(mll), "alpha": alpha.astype(np.float64), "alpha_u": au.astype(np.float64),
            "lam": lam.astype(np.float64), "logdet": float(logdet)}


def predict_loo(fit, y):
    """predict_LOO(Σ, alpha, y) — src/crossvalidation.jl:8-13: σᵢ² = 1 / (Σ⁻¹)ᵢᵢ, μᵢ = yᵢ − αᵢ σᵢ²."""
    U = fit["U"]
    invS = sla.cho_solve((U, False), np.eye(U.shape[0]))
    s2 = 1.0 / np.diag(invS)
    return np.asarray(y, dtype=np.float64) - fit["alpha"] * s2, s2
