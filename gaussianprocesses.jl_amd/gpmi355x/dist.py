"""Row-block sharded / packed exact GP: the Python side is a LAUNCHER only.

The orchestration — block-cyclic ownership, packed stripes, the look-ahead pipeline (next diagonal block factored and the next
panel exchanged under the current trailing update), the distributed solves, predict_f and update_dmll! — lives below the C ABI
(csrc/blocked.cpp behind gpmi_gp_create_blocked; include/gpmi.h).  A blocked handle answers the same gpmi_fit / gpmi_predict /
gpmi_grad as a dense one, so `ShardedGPE` IS `GPE` (gpe.py) created on a blocked handle: update_mll / update_dmll / predict_f
(both full_cov branches) / predict_y / optimize / get_params / set_params are the inherited methods.

What this module supplies is the communicator (one process per GPU, every rank makes the same calls):
    rccl_comm(ctx)          RCCL opened by libgpmi itself (gpmi_comm_create_rccl); the 128-byte unique id travels through
                            whatever group the launcher already has (torch.distributed here: any backend)
    TorchDistComm(group)    collectives delegated to torch.distributed through gpmi_comm_callbacks: backend "nccl" is RCCL,
                            "gloo" works on device buffers too (two processes on one GPU in tests/test_gpu_dist.py)
and `comm=None` = one rank (packed storage on a single device: SURVEY §8f-3) — or, on a context made with
`Context(devices=[...])`, one rank per device of that in-process group (worker threads + peer copies inside libgpmi: the
single-process multi-GPU form of SURVEY §8(b)'s `gpmi_ctx_create(n_devices 1..8)`).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .gpe import GPE, HIPPDMat


def default_block(n):
    """Rows per distributed block (0 lets the library choose: 1024 from 16 384 points (2048 from 131 072 on one rank), 512 from 4096, 256 below)."""
    e = os.environ.get("GPMI_DIST_WD")
    return int(e) if e else 0


class _DevBytes:
    """raw device memory as a CUDA array (what torch.as_tensor wraps without copying)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class Comm:
    """owner of a gpmi_comm handle"""

    h = None
    rank, world = 0, 1

    def close(self):
        if self.h:
            _lib.load().gpmi_comm_destroy(self.h)
            self.h = None

    def selftest(self, ctx):
        """every collective on small device buffers, verified (gpmi_comm_selftest): raises DeviceError when the transport is broken"""
        ctx.check(_lib.load().gpmi_comm_selftest(ctx.h, self.h))

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class TorchDistComm(Comm):
    """gpmi_comm whose collectives are torch.distributed calls on libgpmi's device buffers, enqueued on the stream libgpmi
    names (so they are ordered with its kernels on the device)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist

        if not torch.cuda.is_available():
            raise _lib.DeviceError("TorchDistComm: torch sees no GPU — import torch BEFORE gpmi355x so that the two share one HIP runtime")

        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.error = None
        T = dict(_lib.GpmiCommCallbacks._fields_)
        self._fns = (T["broadcast"](self._bcast), T["all_gather"](self._gather), T["all_reduce_sum"](self._reduce),
                     T["host_allreduce"](self._host))
        cb = _lib.GpmiCommCallbacks()
        cb.user = None
        cb.broadcast, cb.all_gather, cb.all_reduce_sum, cb.host_allreduce = self._fns
        self._cb = cb
        h = C.c_void_p()
        rc = _lib.load().gpmi_comm_create_callbacks(C.byref(cb), self.rank, self.world, C.byref(h))
        if rc != _lib.GPMI_OK:
            raise _lib.DeviceError(f"gpmi_comm_create_callbacks failed (rc={rc})")
        self.h = h

    # -- helpers --
    def _t(self, ptr, nbytes):
        return self.torch.as_tensor(_DevBytes(ptr, nbytes), device=self.device)

    def _on(self, stream):
        return self.torch.cuda.stream(self.torch.cuda.ExternalStream(int(stream), device=self.device)) if stream else _Null()

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as e:  # noqa: BLE001  (a Python exception must not unwind through the C frames)
            self.error = repr(e)
            return 1

    def _bcast(self, user, buf, nbytes, root, stream):
        def go():
            with self._on(stream):
                src = root if self.group is None else self.dist.get_global_rank(self.group, root)
                self.dist.broadcast(self._t(buf, nbytes), src=src, group=self.group)
        return self._guard(go)

    def _gather(self, user, send, recv, each, stream):
        def go():
            with self._on(stream):
                out, inp = self._t(recv, each * self.world), self._t(send, each)
                if self.backend == "nccl":
                    self.dist.all_gather_into_tensor(out, inp, group=self.group)
                else:  # gloo: the list form (its outputs are the row views of `out`)
                    self.dist.all_gather(list(out.view(self.world, each).unbind(0)), inp, group=self.group)
        return self._guard(go)

    def _reduce(self, user, buf, count, es, stream):
        def go():
            with self._on(stream):
                t = self._t(buf, count * es).view(self.torch.float64 if es == 8 else self.torch.float32)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return self._guard(go)

    def _host(self, user, vals, n, op):
        def go():
            v = np.ctypeslib.as_array(vals, shape=(n,))
            t = self.torch.from_numpy(v.copy())
            if self.backend == "nccl":
                t = t.to(self.device)
            R = self.dist.ReduceOp
            self.dist.all_reduce(t, op=R.SUM if op == 0 else (R.MIN if op == 1 else R.MAX), group=self.group)
            v[:] = t.cpu().numpy()
        return self._guard(go)


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class RcclComm(Comm):
    """gpmi_comm on RCCL opened by libgpmi (no torch on the data path).  `exchange_id(id_bytes_or_None) -> id_bytes` moves the
    128-byte unique id from rank 0 to everybody (default: torch.distributed.broadcast_object_list on the default group)."""

    def __init__(self, ctx, rank, world, exchange_id=None):
        lib = _lib.load()
        self.rank, self.world = int(rank), int(world)
        buf = (C.c_char * 128)()
        if self.rank == 0:
            rc = lib.gpmi_comm_unique_id(buf)
            if rc != _lib.GPMI_OK:
                raise _lib.DeviceError("gpmi_comm_unique_id failed: librccl.so could not be opened")
        ident = bytes(buf) if self.rank == 0 else None
        if exchange_id is None:
            exchange_id = _exchange_id_torch
        ident = exchange_id(ident)
        h = C.c_void_p()
        ctx.check(lib.gpmi_comm_create_rccl(ctx.h, ident, self.rank, self.world, C.byref(h)))
        self.h = h


def _exchange_id_torch(ident):
    import torch.distributed as dist

    box = [ident]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def rccl_comm(ctx):
    """RcclComm for the default torch.distributed group (or a single rank when no group is initialised)."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return RcclComm(ctx, dist.get_rank(), dist.get_world_size())
    except ImportError:
        pass
    return RcclComm(ctx, 0, 1, exchange_id=lambda b: b)


class BlockedPDMat(HIPPDMat):
    """gp.cK of a blocked model: the gpmi_gp handle made by gpmi_gp_create_blocked.  The AbstractPDMat surface is HIPPDMat's — `\\`
    (solve), whiten!, logdet, diag(inv(cK)), diag(cholfactors), cholfactors all answer on a blocked handle (results replicated on
    every rank; cholfactors gathers the n x n factor on the host: for inspection at sizes where that is affordable)."""

    def __init__(self, ctx, x_colmajor, bits, comm=None, block=0, stripe_blocks=0):
        self.ctx, self.bits, self.comm = ctx, bits, comm
        d, n = x_colmajor.shape
        self.dim, self.n = d, n
        h = C.c_void_p()
        ctx.check(_lib.load().gpmi_gp_create_blocked(ctx.h, comm.h if comm is not None else None, bits, d, n, x_colmajor.ctypes.data,
                                                     int(block or 0), int(stripe_blocks or 0), C.byref(h)))
        self.h = h
        br, ns, fb = C.c_int64(), C.c_int32(), C.c_int64()
        ctx.check(_lib.load().gpmi_gp_blocked_info(h, C.byref(br), C.byref(ns), C.byref(fb)))
        self.block_rows, self.nstripes, self.factor_bytes = br.value, ns.value, fb.value


class ShardedGPE(GPE):
    """GPE whose factor is held in block-rows: sharded over the ranks of `comm` and / or packed in stripes (same verbs as GPE)."""

    def __init__(self, x, y, mean=None, kernel=None, logNoise=-2.0, dtype=np.float64, comm=None, ctx=None, block=None, stripe_blocks=None):
        self._blocked = dict(comm=comm, block=block if block else default_block(np.asarray(x).shape[-1]), stripe_blocks=stripe_blocks or 0)
        super().__init__(x, y, mean, kernel, logNoise, dtype=dtype, ctx=ctx)

    def _alloc_cK(self):
        return BlockedPDMat(self.ctx, self.x, self.bits, **self._blocked)

    # layout facts (tests, bench)
    @property
    def WD(self):
        return self.cK.block_rows

    @property
    def nblk(self):
        return (self.nobs + self.WD - 1) // self.WD

    @property
    def nown(self):
        """blocks this rank owns (an in-process device group reports its rank 0)"""
        c = self._blocked["comm"]
        r, g = (c.rank, c.world) if c is not None else (0, len(getattr(self.ctx, "devices", [0])))
        return len(range(r, self.nblk, g))

    @property
    def logdet(self):
        return self.cK.logdet()
