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


class AsyncThreadComm(gd.Comm):
    """G virtual ranks = G threads of one process with NO host synchronisation of the device inside a collective: a rank publishes its
    buffer and an event recorded on the stream libgpmi names, the readers make THAT stream wait for the event and copy, the owners then
    wait for the readers' "done" events — libgpmi's in-process LocalComm (csrc/dev_hip.hip) restated on gpmi_comm_callbacks, so that a
    test can put extra latency in front of a collective (`delay_ms`: a spin kernel enqueued on the stream the collective was given) and
    see whether the factorisation really keeps it off the critical path (tests/test_gpu_dist.py::test_injected_collective_latency...)."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.done = [None] * world

    _cycles_per_ms = None

    @classmethod
    def calibrate(cls, device):
        """torch.cuda._sleep counts device clock ticks whose rate differs between runtimes: measure it once"""
        if cls._cycles_per_ms is None:
            torch.cuda._sleep(1000)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20_000_000
            a.record()
            torch.cuda._sleep(n)
            b.record()
            b.synchronize()
            cls._cycles_per_ms = n / a.elapsed_time(b)
        return cls._cycles_per_ms

    def __init__(self, shared, rank, device=0, delay_ms=0.0):
        self.s, self.rank, self.world = shared, rank, shared.world
        self.device = torch.device("cuda", device)
        self.delay_ms = float(delay_ms)
        self.error = None
        self.delayed = 0          # collectives that carried the injected latency
        self._first_gather = True
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

    def _stream(self, stream):
        return torch.cuda.ExternalStream(int(stream), device=self.device) if stream else torch.cuda.default_stream(self.device)

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

    def _delay(self, st):
        if self.delay_ms > 0:
            with torch.cuda.stream(st):
                torch.cuda._sleep(int(self.delay_ms * self.calibrate(self.device)))
            self.delayed += 1

    def _collective(self, st, mine, copies):
        """publish (tensor, ready event); run copies(got) on st after the owners' events; owners wait for every reader"""
        ev = torch.cuda.Event()
        ev.record(st)
        self.s.slots[self.rank] = (mine, ev)
        self.s.barrier.wait()
        got = list(self.s.slots)
        for q, (_, e) in enumerate(got):
            if q != self.rank:
                st.wait_event(e)
        with torch.cuda.stream(st):
            copies([t for t, _ in got])
        dn = torch.cuda.Event()
        dn.record(st)
        self.s.done[self.rank] = dn
        self.s.barrier.wait()
        for q, e in enumerate(list(self.s.done)):
            if q != self.rank:
                st.wait_event(e)      # my buffer may be overwritten only after everybody has read it
        self.s.barrier.wait()         # the slots are free again

    def _bcast(self, user, buf, nbytes, root, stream):
        def go():
            self.log.append(("bcast", nbytes, root))
            st = self._stream(stream)
            self._delay(st)           # once per factorisation step: the inverse broadcast ...
            self._first_gather = True
            mine = self._t(buf, nbytes)
            self._collective(st, mine, lambda got: mine.copy_(got[root], non_blocking=True) if self.rank != root else None)
        return self._guard(go)

    def _gather(self, user, send, recv, each, stream):
        def go():
            self.log.append(("gather", each))
            st = self._stream(stream)
            if self._first_gather:    # ... and the panel exchange that follows it (its per-group gathers are one exchange)
                self._delay(st)
                self._first_gather = False
            out = self._t(recv, each * self.world)

            def copies(got):
                for q in range(self.world):
                    out[q * each:(q + 1) * each].copy_(got[q], non_blocking=True)
            self._collective(st, self._t(send, each), copies)
        return self._guard(go)

    def _reduce(self, user, buf, count, es, stream):
        def go():
            st = self._stream(stream)
            mine = self._t(buf, count * es).view(torch.float64 if es == 8 else torch.float32)
            with torch.cuda.stream(st):
                tmp = torch.empty((self.world, mine.numel()), dtype=mine.dtype, device=self.device)

            def copies(got):
                for q in range(self.world):
                    tmp[q].copy_(got[q].view(mine.dtype), non_blocking=True)
            self._collective(st, mine, copies)
            with torch.cuda.stream(st):
                mine.copy_(tmp.sum(dim=0))   # after every reader's "done": the same order on every rank
                tmp.record_stream(st)
        return self._guard(go)

    def _host(self, user, vals, n, op):
        def go():
            v = np.ctypeslib.as_array(vals, shape=(n,))
            self.s.slots[self.rank] = v.copy()
            self.s.barrier.wait()
            got = list(self.s.slots)
            r = np.sum(got, axis=0) if op == 0 else (np.min(got, axis=0) if op == 1 else np.max(got, axis=0))
            self.s.barrier.wait()
            v[:] = r
        return self._guard(go)
