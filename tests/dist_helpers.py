"""Test doubles for the sharded path on the GPU (TEST INFRASTRUCTURE — never imported by the product):

LocalThreadComm  an in-process communicator for "virtual ranks" (threads, one gpmi context each) on the single GPU of the
                 test box: the gpmi_comm_callbacks of include/gpmi.h implemented with device-to-device copies between the
                 ranks' buffers (torch views of the raw pointers), rendezvous on a barrier.
The CPU-side stand-in for the device back end lives in tests/hostdev.py / tests/hostdev/host_dev.cpp."""
import ctypes as C
import threading

import numpy as np
import torch

from gpmi355x import _lib
from gpmi355x import dist as gd


class LocalThreadComm(gd.Comm):
    """G virtual ranks = G threads of one process; collectives rendezvous on a barrier and move data with torch copies."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank, device=0):
        self.s, self.rank, self.world = shared, rank, shared.world
        self.device = torch.device("cuda", device)
        self.error = None
        self.log = []
        T = dict(_lib.GpmiCommCallbacks._fields_)
        self._fns = (T["broadcast"](self._bcast), T["all_gather"](self._gather), T["all_reduce_sum"](self._reduce),
                     T["host_allreduce"](self._host))
        cb = _lib.GpmiCommCallbacks()
        cb.user = None
        cb.broadcast, cb.all_gather, cb.all_reduce_sum, cb.host_allreduce = self._fns
        self._cb = cb
        h = C.c_void_p()
        assert _lib.load().gpmi_comm_create_callbacks(C.byref(cb), self.rank, self.world, C.byref(h)) == 0
        self.h = h

    def _t(self, ptr, nbytes):
        return torch.as_tensor(gd._DevBytes(ptr, nbytes), device=self.device)

    def _exchange(self, obj):
        torch.cuda.synchronize(self.device)          # everything the ranks enqueued so far has happened
        self.s.slots[self.rank] = obj
        self.s.barrier.wait()
        got = list(self.s.slots)
        return got

    def _done(self):
        torch.cuda.synchronize(self.device)
        self.s.barrier.wait()

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as e:  # noqa: BLE001
            self.error = repr(e)
            try:
                self.s.barrier.abort()
            except Exception:  # noqa: BLE001
                pass
            return 1

    def _bcast(self, user, buf, nbytes, root, stream):
        def go():
            self.log.append(("bcast", nbytes, root))
            mine = self._t(buf, nbytes)
            got = self._exchange(mine)
            if self.rank != root:
                mine.copy_(got[root])
            self._done()
        return self._guard(go)

    def _gather(self, user, send, recv, each, stream):
        def go():
            self.log.append(("gather", each))
            got = self._exchange(self._t(send, each))
            out = self._t(recv, each * self.world)
            for q in range(self.world):
                out[q * each:(q + 1) * each].copy_(got[q])
            self._done()
        return self._guard(go)

    def _reduce(self, user, buf, count, es, stream):
        def go():
            mine = self._t(buf, count * es).view(torch.float64 if es == 8 else torch.float32)
            torch.cuda.synchronize(self.device)  # BEFORE the clone: libgpmi's producers run on its own (non-blocking) streams
            got = self._exchange(mine.clone())
            tot = got[0].clone()
            for g in got[1:]:
                tot += g
            mine.copy_(tot)
            self._done()
        return self._guard(go)

    def _host(self, user, vals, n, op):
        def go():
            v = np.ctypeslib.as_array(vals, shape=(n,))
            got = self._exchange(v.copy())
            r = np.sum(got, axis=0) if op == 0 else (np.min(got, axis=0) if op == 1 else np.max(got, axis=0))
            self._done()
            v[:] = r
        return self._guard(go)

