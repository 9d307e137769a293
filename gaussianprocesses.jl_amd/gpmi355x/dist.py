"""Row-block sharded exact GP: one process per GPU, RCCL over xGMI for the panel exchange.

The factorisation behind `update_mll!` (src/GPE.jl:202-212 → make_posdef!, src/GP.jl:101-112) is the only
coupled part of the path; `cov!` shards trivially (every rank generates its own block-rows of K from a
replicated x).  Layout (DESIGN.md "Row-block sharding"), round 2 — the two-level factorisation of the
single-GPU path (csrc/chol.h) with the super-panel as the distributed block:

  * the row-major lower factor is split into block-rows of WD = 256·2^s rows (1024 by default), dealt round-robin:
    global block b lives on rank b % G at local block b // G (block-cyclic, so the shrinking trailing
    matrix stays balanced).  In the reference's column-major upper factor these are block-COLUMNS;
  * step k: the owner has factored the WD×WD diagonal block and formed its explicit inverse LW_k (gpmi_dev_super_factor);
    LW_k is broadcast (8 MB fp64 at WD = 1024); every rank solves its own rows of block-column k with ONE product
    X ← X·LW_kᵀ; the solved panel rows are ALL-GATHERED (the one real exchange step of the path: (N − k·WD) × WD
    elements per step, N²/2 in total per rank, in N/WD collectives); every rank then applies ONE K = WD trailing update
    to the rows it owns ("staircase" tile shape: a local block only needs columns up to its own global diagonal);
  * look-ahead: the owner of block k+1 updates that diagonal block FIRST, then factors and inverts it on the context's
    side stream UNDER its share of update k (gpmi_dev_side_begin / _end / _join), so the latency-bound chain is off the
    critical path of every rank; libgpmi enqueues on torch's current stream, so collectives and kernels are ordered on
    the device — no host synchronisation inside the step loop;
  * the right-hand side y − μ rides along as one extra row on every rank (forward solve for free);
    logdet is a local sum + all-reduce; the backward solve walks the block-rows in reverse: every rank keeps the partial
    sums of the blocks it owns, the owner of a block receives their total (an all-reduce of WD numbers), solves its block
    and folds it into its own partial sums; α is assembled by one all-reduce at the end;
  * predict: every rank whitens its share of the test points through the replicated LW_k while the panels are
    re-gathered from the stored factor; μ and σ² are gathered at the end.

All device arithmetic is libgpmi's HIP kernels (`gpmi_dev_*`, include/gpmi.h) on buffers this module
allocates as torch tensors so that torch.distributed (backend "nccl" = RCCL) can move them; torch itself
computes nothing but copies.  `comm` and `ops` are injectable: tests run the same orchestration with
world_size-2 gloo on CPU (ops = a NumPy stand-in defined under tests/) and with virtual ranks on one GPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import _lib

LOG2PI = math.log(2.0 * math.pi)


def default_block(n):
    """Rows per distributed block: the super-panel width of the two-level factorisation (GPMI_DIST_WD overrides)."""
    e = os.environ.get("GPMI_DIST_WD")
    if e:
        return int(e)
    return 1024 if n >= 16384 else (512 if n >= 4096 else 256)


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

    def all_reduce_tensor(self, t):
        """in-place sum over the ranks"""
        if self._active():
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

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

    def all_reduce_tensor(self, t):
        pass

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
        self._on_torch_stream = False

    # -- memory --
    def zeros(self, shape):
        return self.torch.zeros(shape, dtype=self.tdtype, device=self.device)

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    # -- streams: inside `with ops.stream_scope():` libgpmi enqueues on torch's current stream, so torch copies, RCCL
    #    collectives and gpmi kernels are ordered on the device and sync() / torch_sync() have nothing to do --
    def stream_scope(self):
        """libgpmi enqueues on torch's current stream (GPMI_DIST_STREAM=default), on a dedicated torch stream made current
        for the scope (own), or keeps its own stream with host synchronisation around every hand-over (host, round 1's
        scheme).  Measured at world 1, N = 50 000: 832 / 836 / 840 ms per step (profiles/r02_sharded_world1.log)."""
        ops = self
        torch = self.torch
        mode = os.environ.get("GPMI_DIST_STREAM", "default")  # own | default | host (host: libgpmi's stream + host syncs, round 1)

        class _Scope:
            def __enter__(self_inner):
                if mode == "host":
                    return
                if mode == "own":
                    if getattr(ops, "_tstream", None) is None:
                        ops._tstream = torch.cuda.Stream(device=ops.device)
                    ops._tstream.wait_stream(torch.cuda.current_stream(ops.device))
                    self_inner.guard = torch.cuda.stream(ops._tstream)
                    self_inner.guard.__enter__()
                st = torch.cuda.current_stream(ops.device)
                ops.ctx.check(ops.lib.gpmi_ctx_set_stream(ops.ctx.h, C.c_void_p(st.cuda_stream), 1))
                ops._on_torch_stream = True

            def __exit__(self_inner, *exc):
                if mode == "host":
                    return False
                ops._on_torch_stream = False
                ops.ctx.check(ops.lib.gpmi_ctx_set_stream(ops.ctx.h, None, 0))  # waits for the stream that is left
                if mode == "own":
                    self_inner.guard.__exit__(*exc)
                    torch.cuda.current_stream(ops.device).wait_stream(ops._tstream)
                return False

        return _Scope()

    def sync(self):  # libgpmi work finished -> torch may touch the buffers
        if not self._on_torch_stream:
            self.ctx.check(self.lib.gpmi_dev_sync(self.ctx.h))

    def torch_sync(self):  # torch / RCCL work finished -> libgpmi may touch the buffers
        if not self._on_torch_stream:
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

    def super_factor(self, blk, linv, invd, lw, pivot_base):
        """in-place Cholesky of the w×w diagonal block + its 64×64 inverses + 1/diag + its explicit inverse lw (w×w)"""
        assert lw.is_contiguous() and linv.is_contiguous()
        self.ctx.check(self.lib.gpmi_dev_super_factor(self.ctx.h, self.bits, self._p(blk), self._ld(blk), blk.shape[0],
                                                      self._p(linv), self._p(invd), self._p(lw), pivot_base))

    def super_rows(self, X, lw):
        """X ← X·LWᵀ"""
        self.ctx.check(self.lib.gpmi_dev_super_rows(self.ctx.h, self.bits, self._p(X), self._ld(X), X.shape[0], X.shape[1], self._p(lw)))

    def update(self, Cv, Av, Bv, mode, g0=0, G=1, nstair_tiles=0, tpb=2):
        self.ctx.check(self.lib.gpmi_dev_update_blocks(self.ctx.h, self.bits, self._p(Cv), self._ld(Cv), self._p(Av), self._ld(Av),
                                                       self._p(Bv), self._ld(Bv), Cv.shape[0], Cv.shape[1], Av.shape[1], mode, g0, G,
                                                       nstair_tiles, tpb, 0))

    def side_begin(self):
        self.ctx.check(self.lib.gpmi_dev_side_begin(self.ctx.h))

    def side_end(self):
        self.ctx.check(self.lib.gpmi_dev_side_end(self.ctx.h))

    def side_join(self):
        self.ctx.check(self.lib.gpmi_dev_side_join(self.ctx.h))

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


class _NoScope:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


# ------------------------------------------------------------------------------------------------
# the sharded model object
# ------------------------------------------------------------------------------------------------
def owned_blocks(rank, world, nblk):
    return list(range(rank, nblk, world))


class _Stripes:
    """The block-rows a rank owns, in stripes of `per` consecutive local blocks.  One stripe (per = None) is the plain
    (rows × npad) matrix.  PACKED storage: stripe s only holds the columns up to the diagonal of its last
    block — the upper triangle is never allocated, N² (1 + 1/S) / 2 elements instead of N² — and every launch over a row
    range becomes one launch per stripe it crosses.  The carried row (y − μ) lives in the last stripe, whose width is npad."""

    def __init__(self, ops, own, WD, npad, per, padded):
        self.WD, self.npad, self.nown = WD, npad, len(own)
        nown = self.nown
        per = nown if (not per or per >= nown) else int(per)
        self.items = []  # (first local block, one past the last, tensor view rows × width)
        starts = list(range(0, nown, per)) if nown else [0]
        for i0 in starts:
            i1 = min(i0 + per, nown)
            last = i1 == nown
            width = npad if (last or per >= nown) else (own[i1 - 1] + 1) * WD
            rows = (i1 - i0) * WD + (8 if last else 0)
            self.items.append((i0, i1, ops.zeros((rows, padded(width)))[:, :width]))
        self.nbytes_rows = sum(t.shape[0] * t.stride(0) for _, _, t in self.items)

    def block(self, i):
        """local block i: WD rows × (its stripe's width)"""
        for i0, i1, t in self.items:
            if i0 <= i < i1:
                return t[(i - i0) * self.WD:(i - i0 + 1) * self.WD]
        raise IndexError(i)

    def carried(self):
        i0, i1, t = self.items[-1]
        return t[(i1 - i0) * self.WD]

    def pieces(self, first_block, carried=True):
        """(rows view, first local block, number of blocks, has the carried row) for the local blocks >= first_block"""
        out = []
        for i0, i1, t in self.items:
            last = i1 == self.nown
            b0 = max(i0, first_block)
            nb = max(0, i1 - b0)
            extra = 1 if (carried and last) else 0
            if nb == 0 and not extra:
                continue
            r0 = (b0 - i0) * self.WD if nb else (i1 - i0) * self.WD
            out.append((t[r0:r0 + nb * self.WD + extra], b0, nb, bool(extra)))
        return out


class ShardedGPE:
    """GPE whose factor is row-block sharded over the ranks of `comm` (same verbs as gpe.GPE)."""

    def __init__(self, x, y, mean, kernel, logNoise=-2.0, dtype=np.float64, comm=None, ops=None, ctx=None, block=None,
                 stripe_blocks=None):
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
        WD = int(block) if block else default_block(self.nobs)
        if WD < 256 or WD % 256 or (WD // 256) & (WD // 256 - 1):
            raise _lib.ArgumentError("the distributed block must be 256 * 2^s rows")
        self.WD = WD
        self.tpb = WD // 128                                   # 128-row tiles per distributed block
        self.npad = (self.nobs + WD - 1) // WD * WD
        self.nblk = self.npad // WD
        self.own = owned_blocks(r, G, self.nblk)
        self.nown = len(self.own)
        o = self.ops
        self.x_dev = o.from_host(self.x.T)                     # n × d row-major, replicated (N·d·s bytes)
        # never a row stride that is a multiple of 4 KiB (a power-of-two stride parks every row of a tile on the same HBM
        # channels: 58 instead of 65 TFLOP/s on the K = 1024 update, as on the single-GPU path's ld): 64 spare columns
        self._ldA = self._padded(self.npad)
        # owned block-rows + the carried row y − μ; stripe_blocks = k: packed storage in stripes of k blocks (_Stripes)
        self.S = _Stripes(o, self.own, WD, self.npad, stripe_blocks, self._padded)
        self.A = self.S.items[0][2] if len(self.S.items) == 1 else None
        need_P = G > 1 or len(self.S.items) > 1
        self._Pfull = o.zeros((self.npad, self._padded(WD))) if need_P else None   # the gathered panel, global row order
        self.P = self._Pfull[:, :WD] if need_P else None
        self.LW = o.zeros((self.nblk, WD, WD))                 # explicit inverse of every diagonal block, replicated
        self.linv = o.zeros((max(self.nown, 1), WD, 64))       # 64×64 inverses of the OWN diagonal blocks (back-substitution)
        self.invd = o.zeros((max(self.nown, 1), WD))
        self.alpha_dev = o.zeros((self.npad,))
        self.alpha = None
        self.mll = float("nan")
        self.target = float("nan")
        self.update_mll()
        self.target = self.mll

    # ---- helpers ---------------------------------------------------------------------------------
    def _padded(self, ncols):
        es = 8 if self.bits == 64 else 4
        return ncols + 64 if (ncols * es) % 4096 == 0 else ncols

    def _scope(self):
        return self.ops.stream_scope() if hasattr(self.ops, "stream_scope") else _NoScope()

    def _n_le(self, q, k):
        """number of blocks owned by rank q with global index <= k"""
        return (k - q) // self.comm.world + 1 if k >= q else 0

    def _blocks_below(self, q, k):
        return [b for b in range(q, self.nblk, self.comm.world) if b > k]

    def _panel_rows(self, k):
        """The solved rows of block-column k below its diagonal block, in GLOBAL row order from row (k+1)·WD on: the B
        operand of update k.  One rank, one stripe: the local rows are already that.  One rank, packed stripes: copied
        into self.P.  Otherwise all-gathered into self.P."""
        o, G, WD = self.ops, self.comm.world, self.WD
        k0 = k * WD
        nle = self._n_le(self.comm.rank, k)
        pcs = self.S.pieces(nle, carried=False)
        if G == 1:
            if len(self.S.items) == 1:
                return self.A[nle * WD:self.nown * WD, k0:k0 + WD]
            for view, b0, nb, _ in pcs:                          # local block = global block
                self.P[b0 * WD:(b0 + nb) * WD].copy_(view[:, k0:k0 + WD])
            return self.P[k0 + WD:]
        rows = [len(self._blocks_below(q, k)) * WD for q in range(G)]
        import torch

        send = torch.cat([view[:, k0:k0 + WD] for view, _, _, _ in pcs], dim=0).contiguous() if pcs else self.ops.zeros((0, WD))
        o.sync()
        pieces = self.comm.all_gather_rows(send, rows)
        Pv = self._Pfull.view(self.nblk, WD, self._Pfull.shape[1])[:, :, :WD]
        for q in range(G):
            bq = self._blocks_below(q, k)
            if bq:
                Pv[bq] = pieces[q].reshape(len(bq), WD, WD)
        o.torch_sync()
        return self.P[k0 + WD:]

    # ---- update_mll! ------------------------------------------------------------------------------
    def update_mll(self):
        with self._scope():
            return self._update_mll()

    def _update_mll(self):
        o, comm = self.ops, self.comm
        r, G, WD, tpb = comm.rank, comm.world, self.WD, self.tpb
        n, npad, nblk, nown = self.nobs, self.npad, self.nblk, self.nown
        ymu = np.zeros(npad, dtype=self.npdt)
        ymu[:n] = self.y - self.mean.mean(self.x)
        self.kdiag = o.set_kernel(self.kernel, self.dim)
        o.info(reset=True)
        S = self.S
        for i, b in enumerate(self.own):                       # cov! + nugget, own block-rows only (lower tiles)
            blk_rows = S.block(i)
            o.assemble(self.x_dev, n, self.dim, b * WD, self.logNoise, blk_rows, blk_rows.shape[1])
        ymu_dev = o.from_host(ymu)
        o.sync()
        S.carried().copy_(ymu_dev)
        o.torch_sync()
        if r == 0:                                             # the first diagonal block has nothing to hide behind
            o.super_factor(S.block(0)[:, 0:WD], self.linv[0], self.invd[0], self.LW[0], 0)
        for k in range(nblk):
            k0, owner = k * WD, k % G
            o.sync()
            comm.broadcast(self.LW[k], owner)
            o.torch_sync()
            nle = self._n_le(r, k)
            pcs = S.pieces(nle)                                 # owned rows below block k + the carried y row, by stripe
            for view, _, _, _ in pcs:
                o.super_rows(view[:, k0:k0 + WD], self.LW[k])   # X ← X·LW_kᵀ
            k1 = k0 + WD
            if npad - k1 <= 0:
                continue
            B = self._panel_rows(k)
            mine_next = nle < nown and self.own[nle] == k + 1
            look = False
            if mine_next:
                # this rank owns the NEXT diagonal block: its own tiles first, then its factorisation and inverse — on the
                # side stream under the rest of the update (look-ahead) while that update is longer than the chain beside
                # it (~0.4 ms per 256 columns on contended CUs, as csrc/chol.h decides it), in line otherwise
                blk_rows = S.block(nle)
                blk = blk_rows[:, k1:k1 + WD]
                o.update(blk, blk_rows[:, k0:k0 + WD], B[:WD], 1)
                rest_rows = (nown - nle - 1) * WD + 1
                tiles = (rest_rows / 128.0) * ((npad - k1) / 256.0) * (WD / 256.0)   # in 128 x 128 x 256 tile products
                # beside the update the chain takes ~3x its in-line time: look ahead once the update outlasts ~2/3 of that
                look = tiles >= 1200.0 * (WD // 256)
                if look:
                    o.side_begin()
                o.super_factor(blk, self.linv[nle], self.invd[nle], self.LW[k + 1], k1)
                if look:
                    o.side_end()
            for view, b0, nb, extra in pcs:
                skip = 1 if (mine_next and b0 == nle and nb > 0) else 0   # the next diagonal block had its update already
                rows = view[skip * WD:]
                nbl = nb - skip
                if rows.shape[0] == 0:
                    continue
                g0 = (self.own[b0 + skip] - (k + 1)) if nbl > 0 else 0
                o.update(rows[:, k1:], rows[:, k0:k0 + WD], B, 2, g0, G, tpb * nbl, tpb)
            if look:
                o.side_join()
        o.sync()
        # the FIRST failing pivot wins (ranks past it have been factoring garbage), as dpotrf reports it
        mine = o.info()
        info = comm.all_reduce(float(mine) if mine > 0 else 1e18, "min")
        if info < 1e17:
            raise _lib.PosDefException(int(info))
        # logdet = 2 Σ log L_ii: local share + all-reduce
        half = sum(o.logdiag_sum(S.block(i), b * WD) for i, b in enumerate(self.own))
        self.logdet = 2.0 * comm.all_reduce(half, "sum")
        # backward solve L' α = z, block-rows in reverse.  v = this rank's share of z − Σ_{solved blocks} L_b' α_b: rank 0
        # starts from z (replicated: every rank carried y − μ), the others from 0; the owner of block c needs the TOTAL of
        # its WD entries (an all-reduce of WD numbers), solves, and folds L_c' α_c into its own v
        v = S.carried().clone()
        if r != 0:
            v.zero_()
        self.alpha_dev.zero_()
        o.torch_sync()
        for c in reversed(range(nblk)):
            c0, owner = c * WD, c % G
            if G > 1:
                seg = v[c0:c0 + WD].clone()
                o.sync()
                comm.all_reduce_tensor(seg)
                if r == owner:
                    v[c0:c0 + WD].copy_(seg)
                o.torch_sync()
            if r == owner:
                o.bsolve_block(S.block(c // G), c0, self.linv[c // G], v, self.alpha_dev)
        o.sync()
        comm.all_reduce_tensor(self.alpha_dev)                  # every block of α was written by exactly one rank
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
        with self._scope():
            return self._predict_f(xpred)

    def _predict_f(self, xpred):
        """Posterior mean / variance (full_cov=False branch of src/GP.jl:64-79), test points split over ranks."""
        o, comm = self.ops, self.comm
        r, G, WD = comm.rank, comm.world, self.WD
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
        R = o.zeros((max(pr, 1), self._ldA))[:, :npad]
        mu = o.zeros((max(pr, 1),))
        var = o.zeros((max(pr, 1),))
        if pr > 0:
            xs = o.from_host(xp[:, lo:hi].T)
            mx = o.from_host(np.asarray(self.mean.mean(xp[:, lo:hi]), dtype=self.npdt))
            o.torch_sync()
            o.cov_rows(xs, self.x_dev, self.dim, R[:pr], npad)
            o.row_gemv(R[:pr], n, self.alpha_dev, mx, mu)
        for k in range(nblk):
            k0 = k * WD
            if pr > 0:
                o.super_rows(R[:pr, k0:k0 + WD], self.LW[k])   # V_k = R_k·LW_kᵀ
            if npad - (k0 + WD) > 0:
                B = self._panel_rows(k)                         # every rank takes part, with or without test rows
                if pr > 0:
                    o.update(R[:pr, k0 + WD:], R[:pr, k0:k0 + WD], B, 0)
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

    def predict_y(self, xpred):
        """predict_f + the observation noise (src/GPE.jl:408-416; scalar logNoise)"""
        mu, s2 = self.predict_f(xpred)
        return mu, s2 + np.exp(2.0 * float(np.atleast_1d(self.logNoise)[0]))

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
