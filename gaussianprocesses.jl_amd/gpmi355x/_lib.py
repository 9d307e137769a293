"""ctypes binding of libgpmi.so (include/gpmi.h).

This is the same binding a Julia `ccall` shim makes (see ../julia/GPMI355X.jl and
INTEGRATION.md).  There is deliberately NO fallback: if the HIP library is missing
or no gfx950 device is present, importing the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libgpmi.so")

GPMI_OK, GPMI_ENOTPD, GPMI_EARG, GPMI_EDEVICE = 0, 1, 2, 3
PROF_SYRK, PROF_COV, PROF_PANEL, PROF_SOLVE, PROF_PREDICT = range(5)
# phases of one block step of a blocked / sharded factorisation (include/gpmi.h GPMI_PROF_STEP_*)
STEP_PHASES = {"U1": 5, "chain": 6, "broadcast": 7, "U2a": 8, "solve": 9, "gather": 10, "U2b": 11}

# every symbol include/gpmi.h declares (tests/test_abi.py checks header == this list == library)
SYMBOLS = [
    "gpmi_ctx_create", "gpmi_ctx_destroy", "gpmi_ctx_synchronize", "gpmi_last_error", "gpmi_version",
    "gpmi_gp_create", "gpmi_gp_destroy", "gpmi_fit", "gpmi_predict", "gpmi_grad", "gpmi_cov",
    "gpmi_inv_diag", "gpmi_fitc_create", "gpmi_fitc_destroy", "gpmi_fitc_fit", "gpmi_fitc_predict", "gpmi_fitc_alpha_u", "gpmi_fitc_grad",
    "gpmi_solve", "gpmi_whiten", "gpmi_logdet", "gpmi_factor_to_host", "gpmi_factor_diag",
    "gpmi_profile_enable", "gpmi_profile_get", "gpmi_profile_get_bytes", "gpmi_mfma_peak", "gpmi_bench_gemm",
    "gpmi_comm_create_callbacks", "gpmi_comm_unique_id", "gpmi_comm_create_rccl", "gpmi_comm_destroy", "gpmi_comm_selftest", "gpmi_gp_create_blocked",
    "gpmi_gp_blocked_info", "gpmi_update_alpha",
]


class GpmiCommCallbacks(C.Structure):
    """gpmi_comm_callbacks (include/gpmi.h): collectives on device buffers supplied by the host program"""
    _fields_ = [
        ("user", C.c_void_p),
        ("broadcast", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)),
        ("all_gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)),
        ("all_reduce_sum", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)),
        ("host_allreduce", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32)),
    ]


class GpmiKernel(C.Structure):
    _fields_ = [
        ("n_ops", C.c_int32),
        ("ops", C.POINTER(C.c_int32)),
        ("dims_off", C.POINTER(C.c_int32)),
        ("dims", C.POINTER(C.c_int32)),
        ("params", C.POINTER(C.c_double)),
        ("n_params", C.c_int32),
    ]


class PosDefException(ArithmeticError):
    """LinearAlgebra.PosDefException(info) — raised where the reference's cholesky! throws (src/GP.jl:110)."""

    def __init__(self, info):
        super().__init__(f"matrix is not positive definite; Cholesky factorization failed (info={info}).")
        self.info = int(info)


class ArgumentError(ValueError):
    """Julia ArgumentError (src/GPE.jl:42,129; src/GP.jl:65,103; src/kernels/kernels.jl:34)."""


class DeviceError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libgpmi.so and declare prototypes.  Fails loudly when the extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # libgpmi needs no torch (round 3: the sharded / packed orchestration and the RCCL communicator live below the C ABI).  One
    # thing to know when a host program uses BOTH: torch's wheel bundles its own libamdhip64, and whichever HIP runtime is loaded
    # SECOND in a process finds no GPU.  If torch is already imported, libgpmi.so resolves libamdhip64 to the copy torch loaded and
    # the two share one runtime; a program that imports gpmi355x first and torch later must import torch first instead
    # (bench.py, tests/conftest.py and the communicators of gpmi355x.dist that wrap torch.distributed do).
    import sys

    if "torch" in sys.modules:
        try:
            import torch  # noqa: F401  (make sure its shared libraries are mapped before dlopen)
        except Exception as e:  # noqa: BLE001
            import warnings

            warnings.warn(f"gpmi355x: torch is half-imported ({e!r}); loading libgpmi without sharing its HIP runtime")
    lib = C.CDLL(LIB_PATH)
    vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
    lib.gpmi_version.restype = C.c_char_p
    lib.gpmi_last_error.restype = C.c_char_p
    lib.gpmi_last_error.argtypes = [vp]
    lib.gpmi_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    lib.gpmi_ctx_destroy.argtypes = [vp]
    lib.gpmi_ctx_destroy.restype = None
    lib.gpmi_ctx_synchronize.argtypes = [vp]
    lib.gpmi_gp_create.argtypes = [vp, C.c_int, C.c_int, i64, vp, C.POINTER(vp)]
    lib.gpmi_gp_destroy.argtypes = [vp]
    lib.gpmi_gp_destroy.restype = None
    lib.gpmi_fit.argtypes = [vp, C.POINTER(GpmiKernel), C.POINTER(dbl), i64, vp, C.POINTER(dbl), vp, C.POINTER(i64)]
    lib.gpmi_update_alpha.argtypes = [vp, vp, C.POINTER(dbl), vp]
    lib.gpmi_predict.argtypes = [vp, C.POINTER(GpmiKernel), i64, vp, vp, C.c_int, vp, vp]
    lib.gpmi_cov.argtypes = [vp, C.POINTER(GpmiKernel), C.c_int, C.c_int, i64, vp, i64, vp, vp]
    lib.gpmi_fitc_create.argtypes = [vp, C.c_int, C.c_int, i64, vp, i64, vp, C.POINTER(vp)]
    lib.gpmi_fitc_destroy.argtypes = [vp]
    lib.gpmi_fitc_destroy.restype = None
    lib.gpmi_fitc_fit.argtypes = [vp, C.POINTER(GpmiKernel), C.c_double, vp, C.POINTER(dbl), vp, C.POINTER(i64)]
    lib.gpmi_fitc_predict.argtypes = [vp, C.POINTER(GpmiKernel), i64, vp, vp, C.c_int, vp, vp]
    lib.gpmi_fitc_alpha_u.argtypes = [vp, vp]
    lib.gpmi_fitc_grad.argtypes = [vp, C.POINTER(GpmiKernel), C.c_double, C.POINTER(dbl), C.c_int32, C.POINTER(dbl)]
    lib.gpmi_grad.argtypes = [vp, C.POINTER(GpmiKernel), C.POINTER(dbl), i64, C.POINTER(dbl), C.c_int32, C.POINTER(dbl)]
    lib.gpmi_solve.argtypes = [vp, i64, vp]
    lib.gpmi_whiten.argtypes = [vp, i64, vp]
    lib.gpmi_logdet.argtypes = [vp, C.POINTER(dbl)]
    lib.gpmi_inv_diag.argtypes = [vp, vp]
    lib.gpmi_factor_to_host.argtypes = [vp, vp]
    lib.gpmi_factor_diag.argtypes = [vp, vp]
    lib.gpmi_profile_enable.argtypes = [vp, C.c_int]
    lib.gpmi_profile_get.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl)]
    lib.gpmi_profile_get_bytes.argtypes = [vp, C.c_int, C.POINTER(dbl)]
    lib.gpmi_mfma_peak.argtypes = [vp, C.c_int, C.POINTER(dbl)]
    lib.gpmi_bench_gemm.argtypes = [vp, C.c_int, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.POINTER(dbl)]
    ci = C.c_int
    lib.gpmi_comm_create_callbacks.argtypes = [C.POINTER(GpmiCommCallbacks), ci, ci, C.POINTER(vp)]
    lib.gpmi_comm_unique_id.argtypes = [vp]
    lib.gpmi_comm_create_rccl.argtypes = [vp, vp, ci, ci, C.POINTER(vp)]
    lib.gpmi_comm_destroy.argtypes = [vp]
    lib.gpmi_comm_destroy.restype = None
    lib.gpmi_comm_selftest.argtypes = [vp, vp]
    lib.gpmi_gp_create_blocked.argtypes = [vp, vp, ci, ci, i64, vp, i64, ci, C.POINTER(vp)]
    lib.gpmi_gp_blocked_info.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_int32), C.POINTER(i64)]
    _lib = lib
    return lib


class Context:
    """gpmi_ctx: one per process and GPU."""

    _default = {}

    def __init__(self, device=0, devices=None):
        """devices=[i, j, ...]: an in-process DEVICE GROUP (gpmi_ctx_create with n_devices > 1): blocked models created on it
        (gpmi355x.dist.ShardedGPE(..., ctx=ctx) / GP(..., packed=True, ctx=ctx)) are row-block sharded over those devices by
        one worker thread each, joined by peer copies — no launcher, no process group.  Everything else runs on devices[0]."""
        lib = load()
        h = C.c_void_p()
        devs = [int(device)] if devices is None else [int(v) for v in devices]
        ids = (C.c_int * len(devs))(*devs)
        rc = lib.gpmi_ctx_create(len(devs), ids, C.byref(h))
        if rc != GPMI_OK:
            raise DeviceError(
                f"gpmi_ctx_create failed (rc={rc}): no usable gfx950 device(s) {devs}. "
                "libgpmi has no CPU backend by design.")
        self.h = h
        self.device = devs[0]
        self.devices = devs

    @classmethod
    def default(cls, device=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if "GPMI_DEVICE" not in os.environ else int(os.environ["GPMI_DEVICE"])
        if device not in cls._default:
            cls._default[device] = cls(device)
        return cls._default[device]

    def check(self, rc, info=0):
        if rc == GPMI_OK:
            return
        msg = load().gpmi_last_error(self.h).decode()
        if rc == GPMI_ENOTPD:
            raise PosDefException(info)
        if rc == GPMI_EARG:
            raise ArgumentError(msg)
        raise DeviceError(msg)

    def synchronize(self):
        """waits for all work on the context's device(s): the bracket around a timed region"""
        self.check(load().gpmi_ctx_synchronize(self.h))

    def profile_enable(self, on=True, only=None, skip_chain=False, phases_only=False):
        """HIP-event brackets around the profiled launches: every class, one class (only=PROF_SYRK ...), every class but the
        thousands of tiny chain kernels (skip_chain), or only the per-step phases of a blocked factorisation (phases_only)"""
        code = 0 if not on else ((2 + int(only)) if only is not None else (65 if phases_only else (64 if skip_chain else 1)))
        self.check(load().gpmi_profile_enable(self.h, code))

    def profile_get(self, cls_id):
        n, ms, work = C.c_int64(), C.c_double(), C.c_double()
        self.check(load().gpmi_profile_get(self.h, cls_id, C.byref(n), C.byref(ms), C.byref(work)))
        return n.value, ms.value, work.value

    def profile_get_bytes(self, cls_id):
        b = C.c_double()
        self.check(load().gpmi_profile_get_bytes(self.h, cls_id, C.byref(b)))
        return b.value

    def mfma_peak(self, dtype=64):
        out = C.c_double()
        self.check(load().gpmi_mfma_peak(self.h, dtype, C.byref(out)))
        return out.value

    def bench_gemm(self, M, N, K, lower=1, variant=0, iters=5, dtype=64):
        out = C.c_double()
        self.check(load().gpmi_bench_gemm(self.h, dtype, M, N, K, lower, variant, iters, C.byref(out)))
        return out.value

    def close(self):
        if self.h:
            load().gpmi_ctx_destroy(self.h)
            self.h = None


def np_dtype(bits):
    return np.float64 if bits == 64 else np.float32


def colmajor(x, dtype):
    """d x n array -> Fortran-contiguous buffer (what Julia hands to ccall)."""
    return np.asfortranarray(np.asarray(x, dtype=dtype))
