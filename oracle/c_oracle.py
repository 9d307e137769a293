"""ctypes front-end of oracle/cov_oracle.c (TEST INFRASTRUCTURE ONLY).

Also holds the oracle-side flattening of a kernel spec into the gpmi_kernel
postfix descriptor (include/gpmi.h); the product has its own, independent
serialiser (gaussianprocesses.jl_amd/gpmi355x/kernels.py) and
tests/test_descriptor.py checks the two agree.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

OPS = {
    "se_iso": 1, "se_ard": 2, "mat12_iso": 3, "mat12_ard": 4, "mat32_iso": 5, "mat32_ard": 6,
    "mat52_iso": 7, "mat52_ard": 8, "rq_iso": 9, "rq_ard": 10, "noise": 11, "const": 12,
    "sum": 100, "prod": 101,
}


class GpmiKernel(C.Structure):
    _fields_ = [
        ("n_ops", C.c_int32),
        ("ops", C.POINTER(C.c_int32)),
        ("dims_off", C.POINTER(C.c_int32)),
        ("dims", C.POINTER(C.c_int32)),
        ("params", C.POINTER(C.c_double)),
        ("n_params", C.c_int32),
    ]


def flatten(spec, d):
    """spec -> (ops, dims_off, dims, params) following the struct-field storage of
    the reference types (se_iso.jl:28, se_ard.jl:31, mat*_iso.jl:30, rq_iso.jl:33 ...)."""
    ops, dims_off, dims, params = [], [0], [], []

    def rec(s, active):
        name = s[0]
        if name in ("sum", "prod"):
            rec(s[1], active)
            rec(s[2], active)
            ops.append(OPS[name])
            dims_off.append(len(dims))
            return
        if name == "masked":
            sub = list(s[2])
            # nested masks compose: inner indices refer to the outer view
            new_active = sub if active is None else [active[i] for i in sub]
            rec(s[1], new_active)
            return
        if name == "fixed":
            rec(s[1], active)
            return
        ops.append(OPS[name])
        if active is not None:
            dims.extend(int(i) for i in active)
        dims_off.append(len(dims))
        nd = d if active is None else len(active)
        if name in ("noise", "const"):
            params.append(math.exp(2.0 * s[1]))
        elif name.endswith("_iso"):
            ll = float(s[1])
            params.append(math.exp(2.0 * ll) if name in ("se_iso", "rq_iso") else math.exp(ll))
            params.append(math.exp(2.0 * float(s[2])))
            if name == "rq_iso":
                params.append(math.exp(float(s[3])))
        else:
            ll = [float(v) for v in s[1]]
            if len(ll) != nd:
                raise ValueError("ARD length-scale count must match the active dims")
            params.extend(math.exp(-2.0 * v) for v in ll)
            params.append(math.exp(2.0 * float(s[2])))
            if name == "rq_ard":
                params.append(math.exp(float(s[3])))

    rec(spec, None)
    return (np.asarray(ops, np.int32), np.asarray(dims_off, np.int32),
            np.asarray(dims if dims else [0], np.int32), np.asarray(params, np.float64))


def _as_struct(flat):
    ops, dims_off, dims, params = flat
    k = GpmiKernel()
    k.n_ops = len(ops)
    k.ops = ops.ctypes.data_as(C.POINTER(C.c_int32))
    k.dims_off = dims_off.ctypes.data_as(C.POINTER(C.c_int32))
    k.dims = dims.ctypes.data_as(C.POINTER(C.c_int32))
    k.params = params.ctypes.data_as(C.POINTER(C.c_double))
    k.n_params = len(params)
    return k


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.oracle_cov_sym.argtypes = [C.POINTER(GpmiKernel), C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        _lib.oracle_cov_rect.argtypes = [C.POINTER(GpmiKernel), C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        _lib.oracle_assemble.argtypes = [C.POINTER(GpmiKernel), C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return _lib


def _colmajor(x):
    return np.asfortranarray(np.asarray(x, dtype=np.float64))


def cov(spec, X1, X2=None):
    """cov!(cK, k, X1[, X2]) with the scalar reference-order loops."""
    X1 = _colmajor(X1)
    d, n1 = X1.shape
    flat = flatten(spec, d)
    k = _as_struct(flat)
    if X2 is None:
        out = np.empty((n1, n1), order="F")
        lib().oracle_cov_sym(C.byref(k), d, n1, X1.ctypes.data, out.ctypes.data)
        return out
    X2 = _colmajor(X2)
    n2 = X2.shape[1]
    out = np.empty((n1, n2), order="F")
    lib().oracle_cov_rect(C.byref(k), d, n1, X1.ctypes.data, n2, X2.ctypes.data, out.ctypes.data)
    return out


def assemble(spec, x, log_noise):
    """cov! + nugget of update_cK! (no Cholesky)."""
    x = _colmajor(x)
    d, n = x.shape
    flat = flatten(spec, d)  # keep the arrays alive while C holds pointers into them
    k = _as_struct(flat)
    ln = np.atleast_1d(np.asarray(log_noise, dtype=np.float64))
    out = np.empty((n, n), order="F")
    lib().oracle_assemble(C.byref(k), d, n, x.ctypes.data, ln.ctypes.data, len(ln), out.ctypes.data)
    return out
