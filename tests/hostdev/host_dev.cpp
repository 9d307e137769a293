// host_dev.cpp — TEST INFRASTRUCTURE ONLY (never linked into libgpmi.so, never imported by the product).
//
// A host-memory stand-in for the device back end of the blocked exact-GP driver (csrc/dev.h), so that the ORCHESTRATION
// source the product ships (csrc/blocked.cpp: ownership, stripes, staircase bookkeeping, the look-ahead pipeline, the order
// of the collectives, the distributed solves / predict / gradient) runs under world_size-2 gloo without a GPU.  The heavy
// operations are C callbacks implemented by the test in NumPy + the oracle (tests/hostdev.py); streams are the host's
// program order.  fp64 only.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <memory>

#include "../../gaussianprocesses.jl_amd/csrc/blocked.h"
#include "../../gaussianprocesses.jl_amd/csrc/comm_callbacks.h"

extern "C" {
typedef struct hostdev_ops {
    void (*assemble)(const double* x, int64_t n, int d, int64_t row_off, int64_t nrows, double nugget, const double* nvec, double* A, int64_t ld,
                     int64_t ncols);
    void (*cov_rows)(const double* xa, int64_t na, const double* xb, int64_t nb, int d, double* C, int64_t ldc, int64_t ncols_total);
    int64_t (*super_factor)(double* blk, int64_t ld, int64_t w, double* linv, double* invd, double* lw, int64_t pivot_base);
    void (*gemm)(double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int mode,
                 int g0, int G, int nstair, int tpb, int flags);
    void (*bsolve_block)(const double* Lrows, int64_t ld, int64_t c0, int64_t nb, const double* linv, double* z, double* alpha);
    void (*dmll_rect)(const double* xa, int64_t na, const double* xb, int64_t nb, int d, const double* Wt, int64_t ld, int n_hyp, double* out);
    double (*kdiag)(int* n_hyp);
} hostdev_ops;
}

namespace {

using namespace gpmi;

struct HostDev : Dev {
    hostdev_ops ops;
    int64_t info_ = 0;
    explicit HostDev(const hostdev_ops& o) : ops(o) { es = 8; }
    void* alloc(int64_t bytes) override { return malloc((size_t)bytes); }
    void release(void* p) override { free(p); }
    void zero(void* p, int64_t bytes) override { memset(p, 0, (size_t)bytes); }
    void copy2d(void* dst, int64_t dp, const void* src, int64_t sp, int64_t w, int64_t rows) override {
        for (int64_t r = 0; r < rows; ++r) memmove((char*)dst + r * dp, (const char*)src + r * sp, (size_t)w);
    }
    void upload(void* dst, const void* host, int64_t bytes) override { memcpy(dst, host, (size_t)bytes); }
    void download(void* host, const void* src, int64_t bytes) override { memcpy(host, src, (size_t)bytes); }
    void begin_call() override {}
    void use(DevStream) override {}
    void whole_cus(bool) override {}
    DevEvent record() override { return (DevEvent)1; }
    void wait(DevEvent) override {}
    void sync() override {}
    void* native_stream() override { return nullptr; }
    int set_kernel(const gpmi_kernel*, int, double* kdiag, int* n_hyp) override {
        *kdiag = ops.kdiag(n_hyp);
        return GPMI_OK;
    }
    void assemble(const void* x, int64_t n, int d, int64_t row_off, int64_t nrows, double nugget, const double* nvec, void* A, int64_t ld,
                  int64_t ncols) override {
        ops.assemble((const double*)x, n, d, row_off, nrows, nugget, nvec, (double*)A, ld, ncols);
    }
    void cov_rows(const void* xa, int64_t na, const void* xb, int64_t nb, int d, void* C, int64_t ldc, int64_t ncols_total) override {
        ops.cov_rows((const double*)xa, na, (const double*)xb, nb, d, (double*)C, ldc, ncols_total);
    }
    int super_factor(void* blk, int64_t ld, int64_t w, void* linv, void* invd, void* lw, int64_t pivot_base) override {
        if (info_) return GPMI_OK;  // an earlier pivot failed: later work is skipped, like the device kernels
        info_ = ops.super_factor((double*)blk, ld, w, (double*)linv, (double*)invd, (double*)lw, pivot_base);
        return GPMI_OK;
    }
    void gemm(void* C, int64_t ldc, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, DevShape s,
              int flags) override {
        if (info_ || M <= 0 || N <= 0 || K <= 0) return;
        ops.gemm((double*)C, ldc, (const double*)A, lda, (const double*)B, ldb, M, N, K, s.mode, s.g0, s.G, s.nstair, s.tpb, flags);
    }
    void bsolve_block(const void* Lrows, int64_t ld, int64_t c0, int64_t nb, const void* linv, const void*, void* z, void* alpha) override {
        if (info_) return;
        ops.bsolve_block((const double*)Lrows, ld, c0, nb, (const double*)linv, (double*)z, (double*)alpha);
    }
    double logdiag_sum(const void* A, int64_t ld, int64_t nrows, int64_t col_off) override {
        double s = 0;
        for (int64_t i = 0; i < nrows; ++i) s += log(((const double*)A)[i * ld + col_off + i]);
        return s;
    }
    int64_t info(bool reset) override {
        if (reset) info_ = 0;
        return info_;
    }
    void row_gemv(const void* R, int64_t ldr, int64_t P, int64_t n, const void* v, const void* add, void* out) override {
        for (int64_t p = 0; p < P; ++p) {
            double s = ((const double*)add)[p];
            for (int64_t j = 0; j < n; ++j) s += ((const double*)R)[p * ldr + j] * ((const double*)v)[j];
            ((double*)out)[p] = s;
        }
    }
    void row_sumsq_acc(const void* R, int64_t ldr, int64_t P, int64_t n, double* acc) override {
        for (int64_t p = 0; p < P; ++p) {
            double s = 0;
            for (int64_t j = 0; j < n; ++j) s += ((const double*)R)[p * ldr + j] * ((const double*)R)[p * ldr + j];
            acc[p] += s;
        }
    }
    double dot(const void* a, const void* b, int64_t n) override {
        double s = 0;
        for (int64_t i = 0; i < n; ++i) s += ((const double*)a)[i] * ((const double*)b)[i];
        return s;
    }
    void set_identity_rows(void* R, int64_t ldr, int64_t nrows, int64_t col_off) override {
        for (int64_t i = 0; i < nrows; ++i) {
            memset((double*)R + i * ldr, 0, (size_t)ldr * 8);
            ((double*)R)[i * ldr + col_off + i] = 1.0;
        }
    }
    void qblock(void* Wt, int64_t ld, int64_t rows, int64_t cols, const void* ar, const void* ac, double w, bool neg, bool diag, int64_t ntrace,
                double* trace_acc) override {
        double* W = (double*)Wt;
        const double sg = neg ? -1.0 : 1.0;
        double tr = 0;
        for (int64_t i = 0; i < rows; ++i)
            for (int64_t j = 0; j < cols; ++j) {
                const double kinv = sg * W[i * ld + j];
                const double q = ((const double*)ar)[i] * ((const double*)ac)[j] - kinv;
                if (diag && i == j && i < ntrace) tr += q;
                W[i * ld + j] = w * q;
            }
        if (diag && trace_acc) *trace_acc += tr;
    }
    void dmll_rect_acc(const void* xa, int64_t na, const void* xb, int64_t nb, int d, const void* Wt, int64_t ld, int n_hyp, double* out) override {
        ops.dmll_rect((const double*)xa, na, (const double*)xb, nb, d, (const double*)Wt, ld, n_hyp, out);
    }
};

struct HostGP {
    std::unique_ptr<HostDev> dev;
    std::unique_ptr<CallbackComm> comm;
    std::unique_ptr<BlockedGP> gp;
};

}  // namespace

extern "C" {

int hostdev_create(const hostdev_ops* ops, const gpmi_comm_callbacks* cb, int rank, int world, int d, int64_t n, const double* x, int64_t block,
                   int stripe_blocks, void** out) {
    auto* h = new HostGP();
    h->dev.reset(new HostDev(*ops));
    if (cb && world > 1) h->comm.reset(new CallbackComm(*cb, rank, world));
    BlockedOpts o;
    o.block = block;
    o.stripe_blocks = stripe_blocks;
    h->gp.reset(new BlockedGP(h->dev.get(), h->comm.get(), d, n, o));
    const int rc = h->gp->init(x);
    *out = h;
    return rc;
}
void hostdev_destroy(void* p) { delete (HostGP*)p; }
const char* hostdev_error(void* p) { return ((HostGP*)p)->gp->error().c_str(); }
int hostdev_fit(void* p, const double* log_noise, int64_t n_noise, const double* ymu, double* mll, double* alpha, int64_t* info) {
    gpmi_kernel k;
    memset(&k, 0, sizeof(k));
    return ((HostGP*)p)->gp->fit(&k, log_noise, n_noise, ymu, mll, alpha, info);
}
int hostdev_predict(void* p, int64_t P, const double* xpred, const double* mean, int full_cov, double* mu, double* var) {
    gpmi_kernel k;
    memset(&k, 0, sizeof(k));
    return ((HostGP*)p)->gp->predict(&k, P, xpred, mean, full_cov, mu, var);
}
int hostdev_grad(void* p, const double* log_noise, int64_t n_noise, double* dkern, int n_kern, double* dnoise) {
    gpmi_kernel k;
    memset(&k, 0, sizeof(k));
    return ((HostGP*)p)->gp->grad(&k, log_noise, n_noise, dkern, n_kern, dnoise);
}
int hostdev_factor_diag(void* p, double* out) { return ((HostGP*)p)->gp->factor_diag(out); }
int hostdev_solve(void* p, int64_t nrhs, double* b, int backward) { return ((HostGP*)p)->gp->solve(nrhs, b, backward != 0); }
int hostdev_update_alpha(void* p, const double* ymu, double* mll, double* alpha) { return ((HostGP*)p)->gp->update_alpha(ymu, mll, alpha); }
int hostdev_inv_diag(void* p, double* out) { return ((HostGP*)p)->gp->inv_diag(out); }
int hostdev_factor_to_host(void* p, double* U) { return ((HostGP*)p)->gp->factor_to_host(U); }
double hostdev_logdet(void* p) { return ((HostGP*)p)->gp->logdet(); }
int64_t hostdev_block_rows(void* p) { return ((HostGP*)p)->gp->block_rows(); }
int hostdev_nstripes(void* p) { return ((HostGP*)p)->gp->nstripes(); }
int64_t hostdev_stored_bytes(void* p) { return ((HostGP*)p)->gp->stored_bytes(); }
}
