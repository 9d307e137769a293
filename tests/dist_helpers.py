"""Test doubles for the sharded path (TEST INFRASTRUCTURE — never imported by the product):

FakeOps          the `ops` interface of gpmi355x.dist implemented with NumPy/SciPy + the oracle on CPU torch
                 tensors, so that the ORCHESTRATION (ownership, collectives, staircase bookkeeping) can run
                 under world_size-2 gloo without a GPU;
LocalThreadComm  an in-process communicator for "virtual ranks" (threads) so that the real HIP ops can be
                 exercised with G > 1 on the single GPU of the test box.
"""
import threading

import numpy as np
import scipy.linalg as sla
import torch

from oracle import gp_oracle as G


class FakeOps:
    def __init__(self, spec, bits=64):
        self.spec = spec
        self.bits = bits
        self.tdtype = torch.float64 if bits == 64 else torch.float32
        self._info = 0

    def zeros(self, shape):
        return torch.zeros(shape, dtype=self.tdtype)

    def from_host(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def sync(self):
        pass

    def torch_sync(self):
        pass

    def set_kernel(self, kernel, d):
        return float(G._kdiag(self.spec, np.zeros((d, 1)))[0])

    def assemble(self, x_dev, n, d, row_off, log_noise, A_rows, ncols):
        x = x_dev.numpy().T.astype(np.float64)  # d × n
        nrows = A_rows.shape[0]
        out = np.zeros((nrows, ncols))
        na = max(0, min(nrows, n - row_off))
        if na > 0:  # packed stripes ask for fewer than n columns (up to the stripe's last diagonal)
            nc = min(n, ncols)
            out[:na, :nc] = G.cov(self.spec, x[:, row_off:row_off + na], x)[:, :nc]
        nv = np.exp(2.0 * np.atleast_1d(np.asarray(log_noise, dtype=np.float64)))
        for i in range(nrows):
            g = row_off + i
            if i < na:
                out[i, g] += nv[0] if nv.shape[0] == 1 else nv[g]
            elif g < ncols:
                out[i, g] = 1.0
        A_rows.copy_(torch.from_numpy(out).to(self.tdtype))

    def cov_rows(self, xa_dev, xb_dev, d, Cview, ncols_total):
        K = G.cov(self.spec, xa_dev.numpy().T.astype(np.float64), xb_dev.numpy().T.astype(np.float64))
        Cview.zero_()
        Cview[:, :K.shape[1]] = torch.from_numpy(K).to(self.tdtype)

    def update(self, Cv, Av, Bv, mode, g0=0, G=1, nstair_tiles=0, tpb=2):
        if self._info or Cv.shape[0] == 0 or Cv.shape[1] == 0:
            return
        Cv -= Av @ Bv[: Cv.shape[1]].T  # full rectangle: a superset of the staircase, the extra part is never read

    def super_factor(self, blk, linv, invd, lw, pivot_base):
        """gpmi_dev_super_factor: Cholesky of the diagonal block + its explicit inverse (the 64x64 inverses are not
        needed by this stand-in: bsolve_block substitutes against the block itself)"""
        if self._info:
            return
        a = blk.numpy().astype(np.float64)
        a = np.tril(a) + np.tril(a, -1).T
        if not np.all(np.isfinite(a)):
            self._info = pivot_base + 1
            return
        L, info = sla.lapack.dpotrf(a, lower=1, clean=1)
        if info != 0:
            self._info = pivot_base + int(info)
            return
        blk.copy_(torch.from_numpy(L).to(self.tdtype))
        invd.copy_(torch.from_numpy(1.0 / np.diag(L)).to(self.tdtype))
        lw.copy_(torch.from_numpy(sla.solve_triangular(L, np.eye(L.shape[0]), lower=True)).to(self.tdtype))

    def super_rows(self, X, lw):
        if self._info or X.shape[0] == 0:
            return
        X.copy_(X @ lw.T)

    def side_begin(self):
        pass

    def side_end(self):
        pass

    def side_join(self):
        pass

    def bsolve_block(self, Lrows, c0, linv, z, alpha):
        nb = Lrows.shape[0]
        l = np.tril(Lrows[:, c0:c0 + nb].numpy().astype(np.float64))
        try:
            a = sla.solve_triangular(l, z[c0:c0 + nb].numpy().astype(np.float64), lower=True, trans="T", check_finite=False)
        except np.linalg.LinAlgError:
            a = np.full(nb, np.nan)
        alpha[c0:c0 + nb] = torch.from_numpy(a).to(self.tdtype)
        if c0 > 0:
            z[:c0] -= Lrows[:, :c0].T @ alpha[c0:c0 + nb]

    def row_gemv(self, R, n, v, add, out):
        out[: R.shape[0]] = add + R[:, :n] @ v[:n]

    def row_var(self, R, n, kdiag, out):
        out[: R.shape[0]] = torch.clamp(kdiag - (R[:, :n] ** 2).sum(dim=1), min=0.0)

    def logdiag_sum(self, A_rows, col_off):
        nb = A_rows.shape[0]
        return float(torch.log(torch.diagonal(A_rows[:, col_off:col_off + nb]).double()).sum())

    def info(self, reset=False):
        if reset:
            self._info = 0
        return self._info


class LocalThreadComm:
    """G virtual ranks = G threads of one process; collectives rendezvous on a barrier."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s = shared
        self.rank = rank
        self.world = shared.world

    def _exchange(self, obj):
        self.s.slots[self.rank] = obj
        self.s.barrier.wait()
        got = list(self.s.slots)
        self.s.barrier.wait()
        return got

    def broadcast(self, t, src):
        got = self._exchange(t.clone() if self.rank == src else None)
        if self.rank != src:
            t.copy_(got[src])

    def all_gather_rows(self, send, rows_per_rank):
        got = self._exchange(send.clone())
        return [got[q][: rows_per_rank[q]] for q in range(self.world)]

    def all_reduce_tensor(self, t):
        got = self._exchange(t.clone())
        tot = got[0].clone()
        for g in got[1:]:
            tot += g
        t.copy_(tot)

    def all_reduce(self, value, op="sum"):
        got = self._exchange(float(value))
        return sum(got) if op == "sum" else (max(got) if op == "max" else min(got))
