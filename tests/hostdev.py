"""TEST INFRASTRUCTURE ONLY — Python half of the host stand-in for the blocked driver's device back end.

`tests/hostdev/host_dev.cpp` compiles the PRODUCT's orchestration source (gaussianprocesses.jl_amd/csrc/blocked.cpp) with g++
against a host-memory `Dev` whose heavy operations call back into this module (NumPy + the oracle).  The tile shapes the
driver asks for are honoured EXACTLY (a product only touches the 128 x 128 tiles its shape names), so wrong ownership /
staircase / offset bookkeeping in the driver shows up as wrong numbers, not as a harmless superset.

Communicators for the callbacks of include/gpmi.h (gpmi_comm_callbacks), on HOST buffers:
    ThreadComm   G virtual ranks = G threads of this process (barrier rendezvous)
    TorchComm    torch.distributed (gloo), one process per rank
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import scipy.linalg as sla

from oracle import gp_oracle as G

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "hostdev", "host_dev.cpp")
DRIVER = os.path.join(ROOT, "gaussianprocesses.jl_amd", "csrc", "blocked.cpp")
LIB = os.path.join(HERE, "hostdev", "_build", "libhostdev.so")

i64, dbl, vp, ci = C.c_int64, C.c_double, C.c_void_p, C.c_int
pd = C.POINTER(C.c_double)


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    deps = [SRC, DRIVER] + [os.path.join(ROOT, "gaussianprocesses.jl_amd", "csrc", f) for f in ("blocked.h", "dev.h", "comm_callbacks.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(f) for f in deps):
        return LIB
    tmp = LIB + f".{os.getpid()}.tmp"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", SRC, DRIVER, "-o", tmp])
    os.replace(tmp, LIB)
    return LIB


class CommCallbacks(C.Structure):
    _fields_ = [
        ("user", vp),
        ("broadcast", C.CFUNCTYPE(ci, vp, vp, i64, ci, vp)),
        ("all_gather", C.CFUNCTYPE(ci, vp, vp, vp, i64, vp)),
        ("all_reduce_sum", C.CFUNCTYPE(ci, vp, vp, i64, ci, vp)),
        ("host_allreduce", C.CFUNCTYPE(ci, vp, pd, C.c_int32, C.c_int32)),
    ]


class HostOps(C.Structure):
    _fields_ = [
        ("assemble", C.CFUNCTYPE(None, vp, i64, ci, i64, i64, dbl, vp, vp, i64, i64)),
        ("cov_rows", C.CFUNCTYPE(None, vp, i64, vp, i64, ci, vp, i64, i64)),
        ("super_factor", C.CFUNCTYPE(i64, vp, i64, i64, vp, vp, vp, i64)),
        ("gemm", C.CFUNCTYPE(None, vp, i64, vp, i64, vp, i64, i64, i64, i64, ci, ci, ci, ci, ci, ci)),
        ("bsolve_block", C.CFUNCTYPE(None, vp, i64, i64, i64, vp, vp, vp)),
        ("dmll_rect", C.CFUNCTYPE(None, vp, i64, vp, i64, ci, vp, i64, ci, vp)),
        ("kdiag", C.CFUNCTYPE(dbl, C.POINTER(ci))),
    ]


def _vec(ptr, n, dtype=np.float64):
    if n <= 0:
        return np.zeros(0, dtype=dtype)
    return np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype)


def _mat(ptr, rows, cols, ld):
    if rows <= 0 or cols <= 0:
        return np.zeros((max(rows, 0), max(cols, 0)))
    flat = _vec(ptr, (rows - 1) * ld + cols)
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(ld * 8, 8))


def tile_cmax(mode, g0, Gw, nstair, tpb, ti, ntn):
    """last 128-column tile that tile-row ti keeps (csrc/tile_order.h)"""
    c = ntn - 1
    if mode == 1:
        c = min(c, ti + g0)
    elif mode == 2 and ti < nstair:
        c = min(c, tpb * (g0 + (ti // tpb) * Gw) + ti % tpb)
    return c


class Ops:
    """the callbacks of one host GP (the oracle kernel `spec` is evaluated in fp64)"""

    def __init__(self, spec):
        self.spec = spec
        self.calls = {"gemm": 0}
        o = HostOps()
        T = dict(HostOps._fields_)
        self._keep = []
        for name in ("assemble", "cov_rows", "super_factor", "gemm", "bsolve_block", "dmll_rect", "kdiag"):
            fn = T[name](getattr(self, "_" + name))
            self._keep.append(fn)
            setattr(o, name, fn)
        self.struct = o

    def _kdiag(self, n_hyp):
        n_hyp[0] = G.num_params(self.spec)
        return float(G._kdiag(self.spec, np.zeros((self._d, 1)))[0])

    def _assemble(self, x, n, d, row_off, nrows, nugget, nvec, A, ld, ncols):
        self._d = d
        X = _mat(x, n, d, d).T  # d x n
        out = _mat(A, nrows, ncols, ld)
        out[:] = 0.0
        na = max(0, min(nrows, n - row_off))
        nc = min(n, ncols)
        if na > 0:
            out[:na, :nc] = G.cov(self.spec, X[:, row_off:row_off + na], X)[:, :nc]
        nv = _vec(nvec, n) if nvec else None
        for i in range(nrows):
            g = row_off + i
            if g >= ncols:
                continue
            if i < na:
                out[i, g] += nugget if nv is None else nv[g]
            else:
                out[i, g] = 1.0

    def _cov_rows(self, xa, na, xb, nb, d, Cp, ldc, ncols_total):
        out = _mat(Cp, na, ncols_total, ldc)
        out[:] = 0.0
        out[:, :nb] = G.cov(self.spec, _mat(xa, na, d, d).T, _mat(xb, nb, d, d).T)

    def _super_factor(self, blk, ld, w, linv, invd, lw, pivot_base):
        a = _mat(blk, w, w, ld)
        s = np.tril(a) + np.tril(a, -1).T
        if not np.all(np.isfinite(s)):
            return pivot_base + 1
        L, info = sla.lapack.dpotrf(s, lower=1, clean=1)
        if info != 0:
            return pivot_base + int(info)
        a[:] = L
        _vec(invd, w)[:] = 1.0 / np.diag(L)
        _mat(lw, w, w, w)[:] = np.tril(sla.solve_triangular(L, np.eye(w), lower=True))
        li = _vec(linv, w * 64).reshape(w // 64, 64, 64)
        for j in range(w // 64):
            li[j] = np.tril(sla.solve_triangular(L[64 * j:64 * j + 64, 64 * j:64 * j + 64], np.eye(64), lower=True))
        return 0

    def _gemm(self, Cp, ldc, Ap, lda, Bp, ldb, M, N, K, mode, g0, Gw, nstair, tpb, flags):
        self.calls["gemm"] += 1
        Cv, Av, Bv = _mat(Cp, M, N, ldc), _mat(Ap, M, K, lda), _mat(Bp, N, K, ldb)
        prod = Av @ Bv.T
        over, neg = bool(flags & 1), bool(flags & 256)
        ntn = (N + 127) // 128
        for ti in range((M + 127) // 128):
            r0, r1 = ti * 128, min(M, ti * 128 + 128)
            ce = min(N, (tile_cmax(mode, g0, Gw, nstair, tpb, ti, ntn) + 1) * 128)
            if ce <= 0:
                continue
            if over:
                Cv[r0:r1, :ce] = -prod[r0:r1, :ce] if neg else prod[r0:r1, :ce]
            else:
                Cv[r0:r1, :ce] -= prod[r0:r1, :ce]

    def _bsolve_block(self, Lrows, ld, c0, nb, linv, z, alpha):
        Lr = _mat(Lrows, nb, c0 + nb, ld)
        zz, al = _vec(z, c0 + nb), _vec(alpha, c0 + nb)
        a = sla.solve_triangular(np.tril(Lr[:, c0:c0 + nb]), zz[c0:c0 + nb], lower=True, trans="T", check_finite=False)
        al[c0:c0 + nb] = a
        if c0 > 0:
            zz[:c0] -= Lr[:, :c0].T @ a

    def _dmll_rect(self, xa, na, xb, nb, d, Wt, ld, n_hyp, out):
        W = _mat(Wt, na, nb, ld)
        _, dK, _ = G.grad_cov_rect(self.spec, _mat(xa, na, d, d).T, _mat(xb, nb, d, d).T)
        o = _vec(out, n_hyp)
        for p in range(n_hyp):
            o[p] += float(np.sum(W * dK[p]))


# ------------------------------------------------------------------------------------------------------------------------
# communicators on host buffers
# ------------------------------------------------------------------------------------------------------------------------
class _CommBase:
    def callbacks(self):
        cb = CommCallbacks()
        T = dict(CommCallbacks._fields_)
        self._keep = [T["broadcast"](self._bcast), T["all_gather"](self._gather), T["all_reduce_sum"](self._reduce),
                      T["host_allreduce"](self._host)]
        cb.user = None
        cb.broadcast, cb.all_gather, cb.all_reduce_sum, cb.host_allreduce = self._keep
        return cb

    def _guard(self, fn, *a):
        try:
            fn(*a)
            return 0
        except BaseException as e:  # noqa: BLE001
            self.error = repr(e)
            return 1


class ThreadComm(_CommBase):
    """virtual ranks: threads of one process"""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world
        self.log = []

    def _exchange(self, obj):
        self.s.slots[self.rank] = obj
        self.s.barrier.wait()
        got = list(self.s.slots)
        self.s.barrier.wait()
        return got

    def _bcast(self, user, buf, nbytes, root, stream):
        def go():
            self.log.append(("bcast", nbytes, root))
            v = _vec(buf, nbytes, np.uint8)
            got = self._exchange(v.copy() if self.rank == root else None)
            if self.rank != root:
                v[:] = got[root]
        return self._guard(go)

    def _gather(self, user, send, recv, each, stream):
        def go():
            self.log.append(("gather", each))
            got = self._exchange(_vec(send, each, np.uint8).copy())
            out = _vec(recv, each * self.world, np.uint8)
            for q in range(self.world):
                out[q * each:(q + 1) * each] = got[q]
        return self._guard(go)

    def _reduce(self, user, buf, count, es, stream):
        def go():
            v = _vec(buf, count, np.float64 if es == 8 else np.float32)
            got = self._exchange(v.copy())
            v[:] = np.sum(got, axis=0)
        return self._guard(go)

    def _host(self, user, vals, n, op):
        def go():
            v = np.ctypeslib.as_array(vals, shape=(n,))
            got = self._exchange(v.copy())
            v[:] = np.sum(got, axis=0) if op == 0 else (np.min(got, axis=0) if op == 1 else np.max(got, axis=0))
        return self._guard(go)


class TorchComm(_CommBase):
    """torch.distributed (gloo) on host buffers: the real collectives, one process per rank"""

    def __init__(self):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def _t(self, ptr, n, dtype):
        return self.torch.from_numpy(_vec(ptr, n, dtype))

    def _bcast(self, user, buf, nbytes, root, stream):
        return self._guard(lambda: self.dist.broadcast(self._t(buf, nbytes, np.uint8), src=root))

    def _gather(self, user, send, recv, each, stream):
        def go():
            out = self._t(recv, each * self.world, np.uint8)
            self.dist.all_gather_into_tensor(out, self._t(send, each, np.uint8).clone())
        return self._guard(go)

    def _reduce(self, user, buf, count, es, stream):
        return self._guard(lambda: self.dist.all_reduce(self._t(buf, count, np.float64 if es == 8 else np.float32)))

    def _host(self, user, vals, n, op):
        def go():
            v = np.ctypeslib.as_array(vals, shape=(n,))
            t = self.torch.from_numpy(v)
            R = self.dist.ReduceOp
            self.dist.all_reduce(t, op=R.SUM if op == 0 else (R.MIN if op == 1 else R.MAX))
        return self._guard(go)


# ------------------------------------------------------------------------------------------------------------------------
class PosDef(Exception):
    def __init__(self, info):
        super().__init__(f"not positive definite (info={info})")
        self.info = info


class HostBlockedGP:
    """The blocked driver on host memory: same verbs as the product's GPE on a blocked handle."""

    def __init__(self, spec, x, y, log_noise, mean_const=0.0, comm=None, block=0, stripe_blocks=0):
        self.lib = C.CDLL(build())
        L = self.lib
        L.hostdev_create.argtypes = [C.POINTER(HostOps), C.POINTER(CommCallbacks), ci, ci, ci, i64, vp, i64, ci, C.POINTER(vp)]
        L.hostdev_destroy.argtypes = [vp]
        L.hostdev_error.restype = C.c_char_p
        L.hostdev_error.argtypes = [vp]
        L.hostdev_fit.argtypes = [vp, pd, i64, vp, pd, vp, C.POINTER(i64)]
        L.hostdev_predict.argtypes = [vp, i64, vp, vp, ci, vp, vp]
        L.hostdev_grad.argtypes = [vp, pd, i64, pd, ci, pd]
        L.hostdev_factor_diag.argtypes = [vp, vp]
        L.hostdev_solve.argtypes = [vp, i64, vp, ci]
        L.hostdev_inv_diag.argtypes = [vp, vp]
        L.hostdev_update_alpha.argtypes = [vp, vp, pd, vp]
        L.hostdev_factor_to_host.argtypes = [vp, vp]
        L.hostdev_logdet.restype = dbl
        L.hostdev_logdet.argtypes = [vp]
        L.hostdev_block_rows.restype = i64
        L.hostdev_block_rows.argtypes = [vp]
        L.hostdev_nstripes.argtypes = [vp]
        L.hostdev_stored_bytes.restype = i64
        L.hostdev_stored_bytes.argtypes = [vp]
        self.spec, self.ops = spec, Ops(spec)
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        self.d, self.n = self.x.shape
        self.ops._d = self.d
        self.log_noise, self.mean_const, self.comm = log_noise, mean_const, comm
        xr = np.ascontiguousarray(self.x.T)
        self._cb = comm.callbacks() if comm is not None else None
        h = vp()
        rc = L.hostdev_create(C.byref(self.ops.struct), C.byref(self._cb) if self._cb is not None else None,
                              comm.rank if comm else 0, comm.world if comm else 1, self.d, self.n, xr.ctypes.data, block, stripe_blocks, C.byref(h))
        self.h = h
        self._check(rc)
        self.update_mll()

    def _check(self, rc, info=0):
        if rc == 0:
            return
        if rc == 1:
            raise PosDef(info)
        raise RuntimeError(f"rc={rc}: {self.lib.hostdev_error(self.h).decode()}")

    def set_spec(self, spec, log_noise=None):
        self.spec = self.ops.spec = spec
        if log_noise is not None:
            self.log_noise = log_noise

    def update_mll(self):
        ln = np.atleast_1d(np.asarray(self.log_noise, dtype=np.float64))
        ymu = np.ascontiguousarray(self.y - self.mean_const)
        mll, info = dbl(), i64()
        alpha = np.empty(self.n)
        rc = self.lib.hostdev_fit(self.h, ln.ctypes.data_as(pd), len(ln), ymu.ctypes.data, C.byref(mll), alpha.ctypes.data, C.byref(info))
        self._check(rc, info.value)
        self.mll, self.alpha, self.logdet = mll.value, alpha, self.lib.hostdev_logdet(self.h)
        return self

    def update_alpha(self, mean_const):
        """update_mll!(kern = false, noise = false) after a change of the mean (GPE.jl:203-211): the factor is kept"""
        self.mean_const = mean_const
        ymu = np.ascontiguousarray(self.y - self.mean_const)
        mll = dbl()
        alpha = np.empty(self.n)
        self._check(self.lib.hostdev_update_alpha(self.h, ymu.ctypes.data, C.byref(mll), alpha.ctypes.data))
        self.mll, self.alpha = mll.value, alpha
        return self

    def predict_f(self, xs, full_cov=False):
        xs = np.asarray(xs, dtype=np.float64)
        P = xs.shape[1]
        xr = np.ascontiguousarray(xs.T)
        mean = np.full(P, float(self.mean_const))
        mu = np.empty(P)
        var = np.empty((P, P)) if full_cov else np.empty(P)
        self._check(self.lib.hostdev_predict(self.h, P, xr.ctypes.data, mean.ctypes.data, 1 if full_cov else 0, mu.ctypes.data, var.ctypes.data))
        return mu, var

    def update_dmll(self):
        nk = G.num_params(self.spec)
        dk = np.empty(nk)
        dn = dbl()
        ln = np.atleast_1d(np.asarray(self.log_noise, dtype=np.float64))
        self._check(self.lib.hostdev_grad(self.h, ln.ctypes.data_as(pd), len(ln), dk.ctypes.data_as(pd), nk, C.byref(dn)))
        self.dkern, self.dnoise = dk, dn.value
        return self

    def factor_diag(self):
        out = np.empty(self.n)
        self._check(self.lib.hostdev_factor_diag(self.h, out.ctypes.data))
        return out

    # AbstractPDMat surface of the blocked handle (b: n or n x nrhs, column-major like Julia's)
    def _rhs(self, b):
        b = np.array(b, dtype=np.float64, order="F", copy=True)
        assert b.shape[0] == self.n
        return b, (1 if b.ndim == 1 else b.shape[1])

    def solve(self, b):
        b, nrhs = self._rhs(b)
        self._check(self.lib.hostdev_solve(self.h, nrhs, b.ctypes.data, 1))
        return b

    def whiten(self, b):
        b, nrhs = self._rhs(b)
        self._check(self.lib.hostdev_solve(self.h, nrhs, b.ctypes.data, 0))
        return b

    def inv_diag(self):
        out = np.empty(self.n)
        self._check(self.lib.hostdev_inv_diag(self.h, out.ctypes.data))
        return out

    def cholfactors(self):
        U = np.empty((self.n, self.n), order="F")
        self._check(self.lib.hostdev_factor_to_host(self.h, U.ctypes.data))
        return U

    @property
    def block_rows(self):
        return self.lib.hostdev_block_rows(self.h)

    @property
    def nstripes(self):
        return self.lib.hostdev_nstripes(self.h)

    @property
    def stored_bytes(self):
        return self.lib.hostdev_stored_bytes(self.h)

    def close(self):
        if self.h:
            self.lib.hostdev_destroy(self.h)
            self.h = None
