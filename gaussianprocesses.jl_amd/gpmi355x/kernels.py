"""Kernel type hierarchy of GaussianProcesses.jl, host side.

Same names, constructor arguments (log scale), stored fields (transformed) and
parameter ordering as the reference so that `get_params`/`set_params!` vectors are
interchangeable:
    SEIso(ll, lσ)      src/kernels/se_iso.jl:28-37     fields ℓ2, σ2
    SEArd(ll[], lσ)    src/kernels/se_ard.jl:31-41     fields iℓ2[], σ2
    Mat12/32/52Iso     src/kernels/mat*_iso.jl:30      fields ℓ, σ2
    Mat12/32/52Ard     src/kernels/mat*_ard.jl:31      fields iℓ2[], σ2
    RQIso(ll, lσ, lα)  src/kernels/rq_iso.jl:33        fields ℓ2, σ2, α
    RQArd(ll[],lσ,lα)  src/kernels/rq_ard.jl:34        fields iℓ2[], σ2, α
    Noise(lσ)          src/kernels/noise.jl:27         field σ2
    Const(lσ)          src/kernels/const.jl:25         field σ2
    SumKernel / ProdKernel (`+`, `*`)  src/kernels/sum_kernel.jl, prod_kernel.jl, pair_kernel.jl:14-24
    Masked(k, active_dims)             src/kernels/masked_kernel.jl:13-24  (0-based dims here)
    FixedKernel(k, free) / fix(k, …)   src/kernels/fixed_kernel.jl

No covariance arithmetic lives here: a kernel only knows how to flatten itself into
the gpmi_kernel postfix descriptor (include/gpmi.h) that the HIP path evaluates.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib

OP = {
    "SEIso": 1, "SEArd": 2, "Mat12Iso": 3, "Mat12Ard": 4, "Mat32Iso": 5, "Mat32Ard": 6,
    "Mat52Iso": 7, "Mat52Ard": 8, "RQIso": 9, "RQArd": 10, "Noise": 11, "Const": 12,
}
OP_SUM, OP_PROD = 100, 101


class Kernel:
    def __add__(self, other):  # sum_kernel.jl:71
        return SumKernel(self, other)

    def __mul__(self, other):  # prod_kernel.jl:71
        return ProdKernel(self, other)

    # -- parameter interface (log scale), overridden by subclasses --
    def get_params(self):
        raise NotImplementedError

    def set_params(self, hyp):
        raise NotImplementedError

    def num_params(self):
        return len(self.get_params())

    # -- descriptor --
    def _flatten(self, d, active, ops, dims_off, dims, params):
        raise NotImplementedError

    def descriptor(self, d):
        """-> (GpmiKernel struct, keep-alive tuple of the backing arrays)."""
        ops, dims_off, dims, params = [], [0], [], []
        self._flatten(d, None, ops, dims_off, dims, params)
        a_ops = np.asarray(ops, dtype=np.int32)
        a_off = np.asarray(dims_off, dtype=np.int32)
        a_dims = np.asarray(dims if dims else [0], dtype=np.int32)
        a_par = np.asarray(params, dtype=np.float64)
        k = _lib.GpmiKernel()
        k.n_ops = len(ops)
        k.ops = a_ops.ctypes.data_as(C.POINTER(C.c_int32))
        k.dims_off = a_off.ctypes.data_as(C.POINTER(C.c_int32))
        k.dims = a_dims.ctypes.data_as(C.POINTER(C.c_int32))
        k.params = a_par.ctypes.data_as(C.POINTER(C.c_double))
        k.n_params = len(params)
        return k, (a_ops, a_off, a_dims, a_par)

    def flat(self, d):
        ops, dims_off, dims, params = [], [0], [], []
        self._flatten(d, None, ops, dims_off, dims, params)
        return ops, dims_off, dims, params

    def grad_slots(self):
        """For each exposed parameter (get_params order) the index of the wrapped tree's parameter it is —
        identity except under FixedKernel (fixed_kernel.jl:63-66).  The device gradient is produced for the
        full tree; this picks the exposed part."""
        return list(range(self.num_params()))

    def _full_num_params(self):
        return self.num_params()


class _Leaf(Kernel):
    def _emit(self, d, active, ops, dims_off, dims, params, stored):
        ops.append(OP[type(self).__name__])
        if active is not None:
            dims.extend(int(i) for i in active)
        dims_off.append(len(dims))
        params.extend(float(v) for v in stored)


class _Iso(_Leaf):
    _sq = True  # ℓ stored squared (SE/RQ) or plain (Matérn)

    def __init__(self, ll, lsig):
        self.set_params([ll, lsig])

    def set_params(self, hyp):
        if len(hyp) != 2:
            raise _lib.ArgumentError(f"{type(self).__name__} has two parameters, received {len(hyp)}.")
        self.l = math.exp(2.0 * hyp[0]) if self._sq else math.exp(hyp[0])
        self.s2 = math.exp(2.0 * hyp[1])

    def get_params(self):
        return [math.log(self.l) / 2.0 if self._sq else math.log(self.l), math.log(self.s2) / 2.0]

    def _flatten(self, d, active, ops, dims_off, dims, params):
        self._emit(d, active, ops, dims_off, dims, params, [self.l, self.s2])


class _Ard(_Leaf):
    def __init__(self, ll, lsig):
        self.set_params(list(ll) + [lsig])

    def set_params(self, hyp):
        hyp = [float(v) for v in hyp]
        if hasattr(self, "il2") and len(hyp) != self.num_params():
            raise _lib.ArgumentError(f"{type(self).__name__} has {self.num_params()} parameters, received {len(hyp)}.")
        self.il2 = [math.exp(-2.0 * v) for v in hyp[:-1]]
        self.s2 = math.exp(2.0 * hyp[-1])

    def get_params(self):
        return [-math.log(v) / 2.0 for v in self.il2] + [math.log(self.s2) / 2.0]

    def _flatten(self, d, active, ops, dims_off, dims, params):
        nd = d if active is None else len(active)
        if len(self.il2) != nd:
            raise _lib.ArgumentError(f"{type(self).__name__}: {len(self.il2)} length scales for {nd} input dimensions")
        self._emit(d, active, ops, dims_off, dims, params, self.il2 + [self.s2])


class SEIso(_Iso):
    pass


class Mat12Iso(_Iso):
    _sq = False


class Mat32Iso(_Iso):
    _sq = False


class Mat52Iso(_Iso):
    _sq = False


class SEArd(_Ard):
    pass


class Mat12Ard(_Ard):
    pass


class Mat32Ard(_Ard):
    pass


class Mat52Ard(_Ard):
    pass


class RQIso(_Leaf):
    def __init__(self, ll, lsig, lalpha):
        self.set_params([ll, lsig, lalpha])

    def set_params(self, hyp):
        if len(hyp) != 3:
            raise _lib.ArgumentError("Rational Quadratic function has three parameters")
        self.l2, self.s2, self.alpha = math.exp(2.0 * hyp[0]), math.exp(2.0 * hyp[1]), math.exp(hyp[2])

    def get_params(self):
        return [math.log(self.l2) / 2.0, math.log(self.s2) / 2.0, math.log(self.alpha)]

    def _flatten(self, d, active, ops, dims_off, dims, params):
        self._emit(d, active, ops, dims_off, dims, params, [self.l2, self.s2, self.alpha])


class RQArd(_Leaf):
    def __init__(self, ll, lsig, lalpha):
        self.set_params(list(ll) + [lsig, lalpha])

    def set_params(self, hyp):
        hyp = [float(v) for v in hyp]
        if hasattr(self, "il2") and len(hyp) != self.num_params():
            raise _lib.ArgumentError(f"RQArd kernel has {self.num_params()} parameters")
        self.il2 = [math.exp(-2.0 * v) for v in hyp[:-2]]
        self.s2 = math.exp(2.0 * hyp[-2])
        self.alpha = math.exp(hyp[-1])

    def get_params(self):
        return [-math.log(v) / 2.0 for v in self.il2] + [math.log(self.s2) / 2.0, math.log(self.alpha)]

    def _flatten(self, d, active, ops, dims_off, dims, params):
        nd = d if active is None else len(active)
        if len(self.il2) != nd:
            raise _lib.ArgumentError(f"RQArd: {len(self.il2)} length scales for {nd} input dimensions")
        self._emit(d, active, ops, dims_off, dims, params, self.il2 + [self.s2, self.alpha])


class _Scalar(_Leaf):
    def __init__(self, lsig):
        self.set_params([lsig])

    def set_params(self, hyp):
        if len(hyp) != 1:
            raise _lib.ArgumentError(f"{type(self).__name__} kernel has one parameter, received {len(hyp)}.")
        self.s2 = math.exp(2.0 * hyp[0])

    def get_params(self):
        return [math.log(self.s2) / 2.0]

    def _flatten(self, d, active, ops, dims_off, dims, params):
        self._emit(d, active, ops, dims_off, dims, params, [self.s2])


class Noise(_Scalar):
    pass


class Const(_Scalar):
    pass


# shortcut constructors (test/kernels.jl:184-205: SE(…) == SEIso(…) etc.)
def SE(ll, lsig):
    return SEArd(ll, lsig) if np.ndim(ll) else SEIso(ll, lsig)


def Matern(nu, ll, lsig):
    iso = {0.5: Mat12Iso, 1.5: Mat32Iso, 2.5: Mat52Iso}
    ard = {0.5: Mat12Ard, 1.5: Mat32Ard, 2.5: Mat52Ard}
    if nu not in iso:
        raise _lib.ArgumentError("Only Matern 1/2, 3/2 and 5/2 are implementable")
    return ard[nu](ll, lsig) if np.ndim(ll) else iso[nu](ll, lsig)


def RQ(ll, lsig, lalpha):
    return RQArd(ll, lsig, lalpha) if np.ndim(ll) else RQIso(ll, lsig, lalpha)


class _Pair(Kernel):
    _op = None

    def __init__(self, kleft, kright):
        self.kleft, self.kright = kleft, kright

    def get_params(self):  # pair_kernel.jl:15
        return self.kleft.get_params() + self.kright.get_params()

    def set_params(self, hyp):  # pair_kernel.jl:18-24
        npl = self.kleft.num_params()
        if len(hyp) != npl + self.kright.num_params():
            raise _lib.ArgumentError("wrong number of parameters for composite kernel")
        self.kleft.set_params(list(hyp[:npl]))
        self.kright.set_params(list(hyp[npl:]))

    def _flatten(self, d, active, ops, dims_off, dims, params):
        self.kleft._flatten(d, active, ops, dims_off, dims, params)
        self.kright._flatten(d, active, ops, dims_off, dims, params)
        ops.append(self._op)
        dims_off.append(len(dims))

    def grad_slots(self):
        off = self.kleft._full_num_params()
        return self.kleft.grad_slots() + [off + i for i in self.kright.grad_slots()]

    def _full_num_params(self):
        return self.kleft._full_num_params() + self.kright._full_num_params()


class SumKernel(_Pair):
    _op = OP_SUM


class ProdKernel(_Pair):
    _op = OP_PROD


class Masked(Kernel):
    """Masked(kern, active_dims) — masked_kernel.jl:13-24.  active_dims are 0-based."""

    def __init__(self, kernel, active_dims):
        self.kernel = kernel
        self.active_dims = [int(i) for i in active_dims]

    def get_params(self):
        return self.kernel.get_params()

    def set_params(self, hyp):
        self.kernel.set_params(hyp)

    def _flatten(self, d, active, ops, dims_off, dims, params):
        new_active = self.active_dims if active is None else [active[i] for i in self.active_dims]
        for i in new_active:
            if not (0 <= i < d):
                raise _lib.ArgumentError("Masked: active dimension out of range")
        self.kernel._flatten(d, new_active, ops, dims_off, dims, params)

    def grad_slots(self):
        return self.kernel.grad_slots()

    def _full_num_params(self):
        return self.kernel._full_num_params()


class FixedKernel(Kernel):
    """FixedKernel(kernel, free): only parameters whose indices are in `free` are exposed
    (fixed_kernel.jl).  cov is delegated unchanged (fixed_kernel.jl:69)."""

    def __init__(self, kernel, free):
        self.kernel = kernel
        self.free = [int(i) for i in free]

    def get_params(self):
        p = self.kernel.get_params()
        return [p[i] for i in self.free]

    def set_params(self, hyp):
        p = self.kernel.get_params()
        if len(hyp) != len(self.free):
            raise _lib.ArgumentError("FixedKernel: wrong number of free parameters")
        for i, v in zip(self.free, hyp):
            p[i] = float(v)
        self.kernel.set_params(p)

    def _flatten(self, d, active, ops, dims_off, dims, params):
        self.kernel._flatten(d, active, ops, dims_off, dims, params)

    def grad_slots(self):
        inner = self.kernel.grad_slots()
        return [inner[i] for i in self.free]

    def _full_num_params(self):
        return self.kernel._full_num_params()


def fix(kernel, *fixed_indices):
    """fix(k) freezes every parameter; fix(k, i, …) freezes parameters i (0-based)."""
    n = kernel.num_params()
    if not fixed_indices:
        return FixedKernel(kernel, [])
    return FixedKernel(kernel, [i for i in range(n) if i not in fixed_indices])


def from_spec(spec):
    """Build a kernel from the nested-tuple spec used by the tests' shared case list."""
    name = spec[0]
    if name == "sum":
        return from_spec(spec[1]) + from_spec(spec[2])
    if name == "prod":
        return from_spec(spec[1]) * from_spec(spec[2])
    if name == "masked":
        return Masked(from_spec(spec[1]), spec[2])
    if name == "fixed":
        return FixedKernel(from_spec(spec[1]), spec[2])
    table = {
        "se_iso": SEIso, "se_ard": SEArd, "mat12_iso": Mat12Iso, "mat12_ard": Mat12Ard,
        "mat32_iso": Mat32Iso, "mat32_ard": Mat32Ard, "mat52_iso": Mat52Iso, "mat52_ard": Mat52Ard,
        "rq_iso": RQIso, "rq_ard": RQArd, "noise": Noise, "const": Const,
    }
    return table[name](*spec[1:])
