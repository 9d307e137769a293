"""Row-block sharded exact GP: one process per GPU, RCCL over xGMI for the panel exchange.

The factorisation behind `update_mll!` (src/GPE.jl:202-212 → make_posdef!, src/GP.jl:101-112) is the only
coupled part of the path; `cov!` shards trivially (every rank generates its own block-rows of K from a
replicated x).  Layout (DESIGN.md "Row-block sharding"):

  * the row-major lower factor is split into block-rows of NBD = 256 rows, dealt round-robin:
    global block b lives on rank b % G at local block b // G (block-cyclic, so the shrinking trailing
    matrix stays balanced).  In the reference's column-major upper factor these are block-COLUMNS;
  * step k: the owner factors the diagonal block and broadcasts it (512 KB fp64); every rank solves its
    own rows of block-column k against it; the solved panel rows are ALL-GATHERED (the one real exchange
    step of the path: (N - k·NBD) × 256 elements per step, N²/2 elements in total per rank); every rank
    then applies the MFMA trailing update to the rows it owns ("staircase" tile shape: a local block
    only needs columns up to its own global diagonal);
  * the right-hand side y − μ rides along as one extra row on every rank (forward solve for free);
    logdet is a local sum + all-reduce; the backward solve walks the block-rows in reverse, the owner of
    each block doing its 256 rows and broadcasting the running vector;
  * predict: every rank whitens its share of the test points while the panels are re-gathered from the
    stored factor; μ and σ² are gathered at the end.

All device arithmetic is libgpmi's HIP kernels (`gpmi_dev_*`, include/gpmi.h) on buffers this module
allocates as torch tensors so that torch.distributed (backend "nccl" = RCCL) can move them; torch itself
computes nothing but copies.  `comm` and `ops` are injectable: tests run the same orchestration with
world_size-2 gloo on CPU (ops = a NumPy stand-in defined under tests/) and with virtual ranks on one GPU.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib

NBD = 256  # rows per distributed block (= K of the MFMA trailing update)
LOG2PI = math.log(2.0 * math.pi)


# ------------------------------------------------------------------------------------------------
# communicators
# ------------------------------------------------------------------------------------------------
class TorchDistComm:
    """torch.distributed process group (nccl == RCCL on ROCm; gloo for the CPU tests).

    `force` (or GPMI_DIST_FORCE=1) issues the collectives even in a group of one: that is how the RCCL calls themselves —
    tensor placement, dtypes, contiguity — are exercised on the single-GPU test box (tests/test_gpu_dist.py)."""

    def __init__(self, group=None, force=None):
        import os

        import torch
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.force = bool(int(os.environ.get("GPMI_DIST_FORCE", "0"))) if force is None else bool(force)
        # scalars are reduced on the device the backend moves data on
        self._dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    def _active(self):
        return self.world > 1 or self.force

    def broadcast(self, t, src):
        if self._active():
            self.dist.broadcast(t, src=src, group=self.group)

    def all_gather_rows(self, send, rows_per_rank):
        """send: (rows_per_rank[rank] × w) contiguous.  Returns one (rows × w) tensor per rank."""
        import torch

        if not self._active():
            return [send]
        mx = max(rows_per_rank)
        w = send.shape[1]
        buf = torch.zeros((mx, w), dtype=send.dtype, device=send.device)
        buf[: send.shape[0]] = send
        out = torch.empty((self.world, mx, w), dtype=send.dtype, device=send.device)
        self.dist.all_gather_into_tensor(out.view(-1), buf.view(-1), group=self.group)
        return [out[q, : rows_per_rank[q]] for q in range(self.world)]

    def all_reduce(self, value, op="sum"):
        import torch

        if not self._active():
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._dev)
        rop = {"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN}[op]
        self.dist.all_reduce(t, op=rop, group=self.group)
        return float(t.item())


class SingleComm:
    rank, world = 0, 1

    def broadcast(self, t, src):
        pass

    def all_gather_rows(self, send, rows_per_rank):
        return [send]

    def all_reduce(self, value, op="sum"):
        return value


# ------------------------------------------------------------------------------------------------
# device ops: thin wrappers over gpmi_dev_* on torch CUDA tensors
# ------------------------------------------------------------------------------------------------
class DeviceOps:
    def __init__(self, ctx, bits):
        import torch

        self.torch = torch
        self.ctx = ctx
        self.bits = bits
        self.lib = _lib.load()
        self.tdtype = torch.float64 if bits == 64 else torch.float32
        self.device = torch.device("cuda", ctx.device)

    # -- memory --
    def zeros(self, shape):
        return self.torch.zeros(shape, dtype=self.tdtype, device=self.device)

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    # -- stream hand-over between libgpmi's stream and torch's --
    def sync(self):  # libgpmi work finished -> torch may touch the buffers
        self.ctx.check(self.lib.gpmi_dev_sync(self.ctx.h))

    def torch_sync(self):  # torch / RCCL work finished -> libgpmi may touch the buffers
        self.torch.cuda.current_stream(self.device).synchronize()

    @staticmethod
    def _p(t):
        assert t.stride(-1) == 1
        return C.c_void_p(t.data_ptr())

    @staticmethod
    def _ld(t):
        return t.stride(0) if t.dim() == 2 else t.shape[0]

    # -- kernels --
    def set_kernel(self, kernel, d):
        kd, keep = kernel.descriptor(d)
        out = C.c_double()
        self.ctx.check(self.lib.gpmi_dev_set_kernel(self.ctx.h, C.byref(kd), d, C.byref(out)))
        del keep
        return out.value

    def assemble(self, x_dev, n, d, row_off, log_noise, A_rows, ncols):
        ln = np.atleast_1d(np.asarray(log_noise, dtype=np.float64))
        self.ctx.check(self.lib.gpmi_dev_assemble(self.ctx.h, self.bits, d, n, self._p(x_dev), row_off, A_rows.shape[0],
                                                  ln.ctypes.data_as(C.POINTER(C.c_double)), ln.shape[0], self._p(A_rows),
                                                  self._ld(A_rows), ncols))

    def cov_rows(self, xa_dev, xb_dev, d, Cview, ncols_total):
        self.ctx.check(self.lib.gpmi_dev_cov_rows(self.ctx.h, self.bits, d, xa_dev.shape[0], self._p(xa_dev), xb_dev.shape[0],
                                                  self._p(xb_dev), self._p(Cview), self._ld(Cview), ncols_total))

    def potrf_block(self, blk, linv, invd, pivot_base):
        self.ctx.check(self.lib.gpmi_dev_potrf_block(self.ctx.h, self.bits, self._p(blk), self._ld(blk), blk.shape[0],
                                                     self._p(linv), self._p(invd), pivot_base))

    def rows_solve(self, X, L, linv):
        self.ctx.check(self.lib.gpmi_dev_rows_solve(self.ctx.h, self.bits, self._p(X), self._ld(X), X.shape[0], self._p(L),
                                                    self._ld(L), self._p(linv), L.shape[0]))

    def update(self, Cv, Av, Bv, mode, g0=0, G=1, nstair_tiles=0):
        self.ctx.check(self.lib.gpmi_dev_update(self.ctx.h, self.bits, self._p(Cv), self._ld(Cv), self._p(Av), self._ld(Av),
                                                self._p(Bv), self._ld(Bv), Cv.shape[0], Cv.shape[1], Av.shape[1], mode, g0, G,
                                                nstair_tiles))

    def bsolve_block(self, Lrows, c0, linv, z, alpha):
        self.ctx.check(self.lib.gpmi_dev_bsolve_block(self.ctx.h, self.bits, self._p(Lrows), self._ld(Lrows), c0,
                                                      Lrows.shape[0], self._p(linv), self._p(z), self._p(alpha)))

    def row_gemv(self, R, n, v, add, out):
        self.ctx.check(self.lib.gpmi_dev_row_gemv(self.ctx.h, self.bits, self._p(R), self._ld(R), R.shape[0], n, self._p(v),
                                                  self._p(add), self._p(out)))

    def row_var(self, R, n, kdiag, out):
        self.ctx.check(self.lib.gpmi_dev_row_var(self.ctx.h, self.bits, self._p(R), self._ld(R), R.shape[0], n, kdiag,
                                                 self._p(out)))

    def logdiag_sum(self, A_rows, col_off):
        out = C.c_double()
        self.ctx.check(self.lib.gpmi_dev_logdiag_sum(self.ctx.h, self.bits, self._p(A_rows), self._ld(A_rows), A_rows.shape[0],
                                                     col_off, C.byref(out)))
        return out.value

    def info(self, reset=False):
        out = C.c_int64()
        self.ctx.check(self.lib.gpmi_dev_info(self.ctx.h, 1 if reset else 0, C.byref(out)))
        return out.value


# ------------------------------------------------------------------------------------------------
# the sharded model object
# ------------------------------------------------------------------------------------------------
def owned_blocks(rank, world, nblk):
    return list(range(rank, nblk, world))


class ShardedGPE:
    """GPE whose factor is row-block sharded over the ranks of `comm` (same verbs as gpe.GPE)."""

    def __init__(self, x, y, mean, kernel, logNoise=-2.0, dtype=np.float64, comm=None, ops=None, ctx=None):
        from .means import MeanZero

        x = np.asarray(x)
        if x.ndim == 1:
            x = x[None, :]
        y = np.asarray(y, dtype=np.float64)
        if y.ndim != 1 or y.shape[0] != x.shape[1]:
            raise _lib.ArgumentError("Input and output observations must have consistent dimensions.")
        self.mean = mean if mean is not None else MeanZero()
        self.kernel = kernel
        self.logNoise = np.asarray(logNoise, dtype=np.float64).copy() if np.ndim(logNoise) else float(logNoise)
        self.bits = 64 if np.dtype(dtype) == np.float64 else 32
        self.npdt = _lib.np_dtype(self.bits)
        self.comm = comm if comm is not None else SingleComm()
        if ops is None:
            ctx = ctx if ctx is not None else _lib.Context.default()
            ops = DeviceOps(ctx, self.bits)
            if isinstance(self.comm, TorchDistComm):
                self.comm._dev = ops.device
        self.ops = ops
        self.x = np.asarray(x, dtype=self.npdt)
        self.y = y
        self.dim, self.nobs = self.x.shape
        r, G = self.comm.rank, self.comm.world
        self.npad = (self.nobs + NBD - 1) // NBD * NBD
        self.nblk = self.npad // NBD
        self.own = owned_blocks(r, G, self.nblk)
        self.nown = len(self.own)
        o = self.ops
        self.x_dev = o.from_host(self.x.T)                     # n × d row-major, replicated (N·d·s bytes)
        self.A = o.zeros((self.nown * NBD + 8, self.npad))     # owned block-rows; row nown·NBD carries y − μ
        self.P = o.zeros((self.npad, NBD))                     # the gathered panel, global row order
        # broadcast buffer: rows [0,256) L_kk, rows [256,320) the four 64x64 inverses (contiguous), row 320 1/diag
        self.D = o.zeros((NBD + NBD // 4 + 1, NBD))
        self.Dall = o.zeros((self.nblk, NBD + NBD // 4 + 1, NBD))   # every factored diagonal block, replicated
        self.alpha_dev = o.zeros((self.npad,))
        self.alpha = None
        self.mll = float("nan")
        self.target = float("nan")
        self.update_mll()
        self.target = self.mll

    # ---- helpers ---------------------------------------------------------------------------------
    def _n_le(self, q, k):
        """number of blocks owned by rank q with global index <= k"""
        return (k - q) // self.comm.world + 1 if k >= q else 0

    def _blocks_below(self, q, k):
        return [b for b in range(q, self.nblk, self.comm.world) if b > k]

    def _gather_panel(self, k):
        """All-gather the solved rows of block-column k into self.P (global row order)."""
        o, G = self.ops, self.comm.world
        k0 = k * NBD
        lstart = self._n_le(self.comm.rank, k) * NBD
        mloc = self.nown * NBD - lstart
        rows = [len(self._blocks_below(q, k)) * NBD for q in range(G)]
        send = self.A[lstart:lstart + mloc, k0:k0 + NBD].contiguous()
        pieces = self.comm.all_gather_rows(send, rows)
        Pv = self.P.view(self.nblk, NBD, NBD)
        for q in range(G):
            bq = self._blocks_below(q, k)
            if bq:
                Pv[bq] = pieces[q].reshape(len(bq), NBD, NBD)
        o.torch_sync()

    # ---- update_mll! ------------------------------------------------------------------------------
    def update_mll(self):
        o, comm = self.ops, self.comm
        r, G = comm.rank, comm.world
        n, npad, nblk, nown = self.nobs, self.npad, self.nblk, self.nown
        ymu = np.zeros(npad, dtype=self.npdt)
        ymu[:n] = self.y - self.mean.mean(self.x)
        self.kdiag = o.set_kernel(self.kernel, self.dim)
        o.info(reset=True)
        for i, b in enumerate(self.own):                       # cov! + nugget, own block-rows only
            o.assemble(self.x_dev, n, self.dim, b * NBD, self.logNoise, self.A[i * NBD:(i + 1) * NBD], npad)
        ymu_dev = o.from_host(ymu)
        self.A[nown * NBD].copy_(ymu_dev)
        o.torch_sync()
        for k in range(nblk):
            k0, owner = k * NBD, k % G
            if r == owner:
                lk = (k // G) * NBD
                blk = self.A[lk:lk + NBD, k0:k0 + NBD]
                o.potrf_block(blk, self.D[NBD:NBD + NBD // 4], self.D[NBD + NBD // 4], k0)
                o.sync()
                self.D[:NBD].copy_(blk)
                o.torch_sync()
            comm.broadcast(self.D, owner)
            self.Dall[k].copy_(self.D)
            o.torch_sync()
            nle = self._n_le(r, k)
            lstart = nle * NBD
            mtot = nown * NBD - lstart + 1                      # owned rows below + the carried y row
            X = self.A[lstart:lstart + mtot, k0:k0 + NBD]
            o.rows_solve(X, self.Dall[k, :NBD], self.Dall[k, NBD:NBD + NBD // 4])
            o.sync()
            ncols = npad - (k0 + NBD)
            if ncols > 0:
                self._gather_panel(k)
                g0 = (self.own[nle] - (k + 1)) if nle < nown else 0
                o.update(self.A[lstart:lstart + mtot, k0 + NBD:], X, self.P[k0 + NBD:], 2, g0, G, 2 * (nown - nle))
        o.sync()
        # the FIRST failing pivot wins (ranks past it have been factoring garbage), as dpotrf reports it
        mine = o.info()
        info = comm.all_reduce(float(mine) if mine > 0 else 1e18, "min")
        if info < 1e17:
            raise _lib.PosDefException(int(info))
        # logdet = 2 Σ log L_ii: local share + all-reduce
        half = sum(o.logdiag_sum(self.A[i * NBD:(i + 1) * NBD], b * NBD) for i, b in enumerate(self.own))
        self.logdet = 2.0 * comm.all_reduce(half, "sum")
        # backward solve L' α = z, block-rows in reverse; the owner of a block does its 256 rows
        z = self.A[nown * NBD].clone()
        self.alpha_dev.zero_()
        o.torch_sync()
        for c in reversed(range(nblk)):
            c0, owner = c * NBD, c % G
            if r == owner:
                lc = (c // G) * NBD
                o.bsolve_block(self.A[lc:lc + NBD], c0, self.Dall[c, NBD:NBD + NBD // 4], z, self.alpha_dev)
                o.sync()
            if G > 1:
                comm.broadcast(self.alpha_dev[c0:c0 + NBD], owner)
                if c0 > 0:
                    comm.broadcast(z[:c0], owner)
                o.torch_sync()
        self.alpha = self.alpha_dev[:n].cpu().numpy().astype(self.npdt)
        dot = float((ymu_dev[:n].double() * self.alpha_dev[:n].double()).sum().item())
        self.mll = -(dot + self.logdet + LOG2PI * n) / 2.0     # GPE.jl:210
        return self

    def update_target(self):
        self.update_mll()
        self.target = self.mll
        return self

    # ---- predict_f ---------------------------------------------------------------------------------
    def predict_f(self, xpred):
        """Posterior mean / variance (full_cov=False branch of src/GP.jl:64-79), test points split over ranks."""
        o, comm = self.ops, self.comm
        r, G = comm.rank, comm.world
        xp = np.asarray(xpred)
        if xp.ndim == 1:
            xp = xp[None, :]
        if xp.shape[0] != self.dim:
            raise _lib.ArgumentError("Gaussian Process object and input observations do not have consistent dimensions")
        xp = np.asarray(xp, dtype=self.npdt)
        P = xp.shape[1]
        bounds = [P * q // G for q in range(G + 1)]
        lo, hi = bounds[r], bounds[r + 1]
        pr = hi - lo
        n, npad, nblk = self.nobs, self.npad, self.nblk
        o.set_kernel(self.kernel, self.dim)
        R = o.zeros((max(pr, 1), npad))
        mu = o.zeros((max(pr, 1),))
        var = o.zeros((max(pr, 1),))
        if pr > 0:
            xs = o.from_host(xp[:, lo:hi].T)
            mx = o.from_host(np.asarray(self.mean.mean(xp[:, lo:hi]), dtype=self.npdt))
            o.torch_sync()
            o.cov_rows(xs, self.x_dev, self.dim, R[:pr], npad)
            o.row_gemv(R[:pr], n, self.alpha_dev, mx, mu)
        for k in range(nblk):
            k0 = k * NBD
            if pr > 0:
                o.rows_solve(R[:pr, k0:k0 + NBD], self.Dall[k, :NBD], self.Dall[k, NBD:NBD + NBD // 4])
            if npad - (k0 + NBD) > 0:
                o.sync()
                self._gather_panel(k)                           # every rank takes part, with or without test rows
                if pr > 0:
                    o.update(R[:pr, k0 + NBD:], R[:pr, k0:k0 + NBD], self.P[k0 + NBD:], 0)
        if pr > 0:
            o.row_var(R[:pr], npad, self.kdiag, var)
        o.sync()
        both = self.ops.zeros((max(pr, 1), 2))
        both[:, 0] = mu
        both[:, 1] = var
        o.torch_sync()
        pieces = comm.all_gather_rows(both[:pr].contiguous(), [bounds[q + 1] - bounds[q] for q in range(G)])
        import torch

        allp = torch.cat([p for p in pieces if p.shape[0] > 0], dim=0).cpu().numpy()
        return allp[:, 0].astype(self.npdt), allp[:, 1].astype(self.npdt)

    # ---- parameters (same ordering as GPE: [logNoise; mean; kernel], src/GPE.jl:447-512) -------------
    def get_params(self):
        return [float(v) for v in list(np.atleast_1d(self.logNoise)) + list(self.mean.get_params()) + list(self.kernel.get_params())]

    def set_params(self, hyp):
        hyp = [float(v) for v in hyp]
        nn = 1 if np.ndim(self.logNoise) == 0 else len(self.logNoise)
        self.logNoise = hyp[0] if np.ndim(self.logNoise) == 0 else np.asarray(hyp[:nn])
        nm = self.mean.num_params()
        if nm:
            self.mean.set_params(hyp[nn:nn + nm])
        self.kernel.set_params(hyp[nn + nm:])
