// dev.h — what the BLOCKED exact-GP driver (blocked.cpp: packed storage on one device, row-block sharding over several)
// needs from a device and from a communicator.  Plain C++, no HIP: blocked.cpp is compiled into libgpmi.so against the HIP
// implementation (dev_hip.hip) and — for the world_size-2 gloo tests that run without a GPU — into a test library against a
// host stand-in that lives under tests/ (tests/hostdev/).  One orchestration source, two back ends.
//
// All `void*` matrices are DEVICE pointers of the model's element type (es = 8 | 4 bytes), row-major, leading dimensions in
// elements.  Every operation is enqueued on the back end's CURRENT stream (use()); nothing synchronises unless it says so.
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/gpmi.h"

namespace gpmi {

// streams of one factorisation (dev_hip.hip maps them onto the context's streams):
//   DS_MAIN  the context's stream (uploads, assembly, solves, predict)
//   DS_UPD   the trailing updates and panel solves that run BESIDE the chain / the exchange: with reserved compute units
//            the CU-masked update stream (common.h upd_stream), otherwise the main stream with free workgroup slots
//   DS_SIDE  the look-ahead chain (factor + invert the next diagonal block): the CU-masked chain stream
//   DS_COMM  the panel exchange: collectives and their scatter copies (an unmasked stream: RCCL's kernels and the copies
//            land on the compute units the masked update leaves free)
enum DevStream { DS_MAIN = 0, DS_UPD = 1, DS_SIDE = 2, DS_COMM = 3 };

// product flags: the GemmFlags of common.h that the driver uses, by value
enum DevGemmFlags { DG_OVERWRITE = 1, DG_KSTART_ROW = 2, DG_KEND_COL = 8, DG_NEGOUT = 256 };

struct DevShape {  // tile_order.h TileShape, the fields a caller sets
    int mode = 0;  // 0 rectangle, 1 lower (col tile <= row tile + g0), 2 staircase of a block-cyclic shard
    int g0 = 0, G = 1, nstair = 0, tpb = 2;
};

typedef void* DevEvent;

struct Dev {
    int es = 8;  // bytes per element
    std::string err;
    virtual ~Dev() {}
    // ---- memory ----
    virtual void* alloc(int64_t bytes) = 0;  // nullptr when out of memory (err set); contents undefined
    virtual void release(void* p) = 0;
    virtual void zero(void* p, int64_t bytes) = 0;
    virtual void copy2d(void* dst, int64_t dpitch_bytes, const void* src, int64_t spitch_bytes, int64_t width_bytes, int64_t rows) = 0;
    virtual void upload(void* dst, const void* host, int64_t bytes) = 0;    // the host buffer is reusable on return
    virtual void download(void* host, const void* src, int64_t bytes) = 0;  // waits for the current stream, then copies
    // ---- streams ----
    virtual void begin_call() = 0;   // a new API call: every event of the previous one is free again
    virtual void use(DevStream s) = 0;
    // how DS_UPD / DS_SIDE share the chip from now on: whole compute units reserved for the chain and the exchange (the update
    // gives up ~4 % but the chain runs undisturbed and collectives find free CUs) or free workgroup slots beside a full-width
    // update (~1.5 %, the chain crawls: fine while the update is long).  The driver decides per step (blocked.cpp).
    virtual void whole_cus(bool on) = 0;
    // how many ranks share the factorisation (1: nothing but the chain ever wants the compute units an update leaves free — the update
    // may take them back as soon as the chain's workgroups exit; > 1: the collectives' kernels need them too)
    virtual void set_world(int world) { (void)world; }
    virtual DevEvent record() = 0;   // on the current stream
    virtual void wait(DevEvent e) = 0;  // the current stream waits for e
    virtual void sync() = 0;         // host waits for every stream; returns after device errors are collected in err
    virtual void* native_stream() = 0;  // the current stream, as the communicator wants it (hipStream_t; nullptr on the host)
    // measurement: begin / end of phase `cls` (a GPMI_PROF_STEP_* class of include/gpmi.h) on the CURRENT stream — both calls on the same
    // stream; a back end without timers ignores them
    virtual void phase(int cls, bool begin) { (void)cls; (void)begin; }
    // ---- kernel program ----
    virtual int set_kernel(const gpmi_kernel* k, int d, double* kdiag, int* n_hyp) = 0;  // GPMI_* status
    // ---- covariance ----
    // rows [row_off, row_off + nrows) of K + noise into A (lower tiles only, identity padding past n): update_cK!, GPE.jl:169-186
    virtual void assemble(const void* x, int64_t n, int d, int64_t row_off, int64_t nrows, double nugget, const double* nugget_vec_dev,
                          void* A, int64_t ld, int64_t ncols) = 0;
    // C[i][j] = k(xa_i, xb_j), columns >= nb zero-filled up to ncols_total
    virtual void cov_rows(const void* xa, int64_t na, const void* xb, int64_t nb, int d, void* C, int64_t ldc, int64_t ncols_total) = 0;
    // ---- factorisation ----
    // dpotrf of the w x w diagonal block in place (pivots offset by pivot_base), its 64 x 64 inverses, 1 / L_ii and the explicit
    // inverse lw (w x w, leading dimension w, strictly-upper part exactly zero); a failed pivot is latched (info())
    virtual int super_factor(void* blk, int64_t ld, int64_t w, void* linv, void* invd, void* lw, int64_t pivot_base) = 0;
    // C[M x N] (-)= A[M x K] B[N x K]'
    virtual void gemm(void* C, int64_t ldc, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                      DevShape shape, int flags) = 0;
    // alpha[c0 .. c0+nb) = L_cc^-T z[c0 ..) ; z[0 .. c0) -= L[c rows, 0 .. c0)' alpha_c     (lw: the block's explicit inverse,
    // leading dimension nb; linv: its 64 x 64 diagonal inverses)
    virtual void bsolve_block(const void* Lrows, int64_t ld, int64_t c0, int64_t nb, const void* linv, const void* lw, void* z, void* alpha) = 0;
    virtual double logdiag_sum(const void* A, int64_t ld, int64_t nrows, int64_t col_off) = 0;  // sum log A[i][col_off + i] (synchronises)
    virtual int64_t info(bool reset) = 0;                                                        // the not-PD latch (read synchronises)
    // ---- reductions over rows ----
    virtual void row_gemv(const void* R, int64_t ldr, int64_t P, int64_t n, const void* v, const void* add, void* out) = 0;  // out = add + R[:, :n] v
    virtual void row_sumsq_acc(const void* R, int64_t ldr, int64_t P, int64_t n, double* acc) = 0;                             // acc[p] += |R[p, :n]|^2 (double)
    virtual double dot(const void* a, const void* b, int64_t n) = 0;                                                            // synchronises
    // ---- gradient (update_dmll!, GPE.jl:298-324) ----
    virtual void set_identity_rows(void* R, int64_t ldr, int64_t nrows, int64_t col_off) = 0;  // R = 0 except R[i][col_off + i] = 1
    // Wt[i][j] = w (a_rows[i] a_cols[j] - sgn Kinv[i][j]) in place (Kinv holds sgn^-1-signed K^-1 entries); when diag != 0 the
    // block is a diagonal block: *trace_acc += sum_{i < ntrace} (a_i^2 - K^-1_ii)
    virtual void qblock(void* Wt, int64_t ld, int64_t rows, int64_t cols, const void* a_rows, const void* a_cols, double w, bool kinv_negated,
                        bool diag, int64_t ntrace, double* trace_acc) = 0;
    // out[0 .. n_hyp) += sum_ij Wt[i][j] dk(xa_i, xb_j)/dtheta_p   (device accumulators, double)
    virtual void dmll_rect_acc(const void* xa, int64_t na, const void* xb, int64_t nb, int d, const void* Wt, int64_t ld, int n_hyp,
                               double* out) = 0;
};

// Collectives on DEVICE buffers, enqueued on `stream` (the back end's native stream); host_allreduce is a host-side scalar
// reduction (pivot latch, logdet).  rank / world as usual.  Return 0 or a non-zero error.
struct Comm {
    int rank = 0, world = 1;
    virtual ~Comm() {}
    virtual int broadcast(void* buf, int64_t bytes, int root, void* stream) = 0;
    virtual int all_gather(const void* send, void* recv, int64_t bytes_each, void* stream) = 0;  // recv: world x bytes_each
    virtual int all_reduce_sum(void* buf, int64_t count, int es, void* stream) = 0;              // es 8: double, 4: float
    virtual int host_allreduce(double* vals, int n, int op /*0 sum, 1 min, 2 max*/) = 0;
    // a run of collectives on one stream that the transport may fuse into one launch (RCCL: ncclGroupStart / ncclGroupEnd)
    virtual void group_begin() {}
    virtual int group_end() { return 0; }
};

}  // namespace gpmi
