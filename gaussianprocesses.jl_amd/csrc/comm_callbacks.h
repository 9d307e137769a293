// comm_callbacks.h — a Comm (dev.h) whose collectives are C function pointers supplied by the host program:
// gpmi_comm_create_callbacks (include/gpmi.h).  This is how a launcher that already owns a process group plugs it in
// (torch.distributed in gpmi355x/dist.py — RCCL as backend "nccl", gloo in the CPU tests; MPI from Julia).  Plain C++.
#pragma once
#include "dev.h"

namespace gpmi {

struct CallbackComm : Comm {
    gpmi_comm_callbacks cb;
    CallbackComm(const gpmi_comm_callbacks& c, int r, int w) : cb(c) {
        rank = r;
        world = w;
    }
    int broadcast(void* buf, int64_t bytes, int root, void* stream) override { return cb.broadcast(cb.user, buf, bytes, root, stream); }
    int all_gather(const void* send, void* recv, int64_t bytes_each, void* stream) override {
        return cb.all_gather(cb.user, send, recv, bytes_each, stream);
    }
    int all_reduce_sum(void* buf, int64_t count, int es, void* stream) override { return cb.all_reduce_sum(cb.user, buf, count, es, stream); }
    int host_allreduce(double* vals, int n, int op) override { return cb.host_allreduce(cb.user, vals, n, op); }
};

}  // namespace gpmi
