// dev_hip.hip — the gfx950 back end of the blocked exact-GP driver (dev.h / blocked.cpp), the communicators (RCCL opened at
// run time; caller-supplied callbacks) and the C ABI that creates blocked handles (include/gpmi.h: gpmi_gp_create_blocked,
// gpmi_comm_*).  Every Dev operation is one of the launchers of common.h on the context's streams:
//   DS_MAIN  ctx->stream as it was when the call began
//   DS_UPD   the CU-masked update stream (248 CUs) when the context reserves compute units, else the main stream with
//            `lookahead_slots` workgroup slots left free
//   DS_SIDE  the CU-masked chain stream (the 8 reserved CUs), else the high-priority side stream; launches are capped to the
//            free slots (common.h side_cap)
//   DS_COMM  the main stream (idle while UPD / SIDE run a factorisation): RCCL's kernels and the scatter copies land on the
//            reserved CUs; with free slots instead of reserved CUs, the side stream
// Only ONE look-ahead stream set exists at a time (common.h set_lookahead_mode): HIP multiplexes streams onto few hardware queues.
#include <dlfcn.h>
#include <math.h>
#include <string.h>

#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "blocked.h"
#include "chol.h"
#include "comm_callbacks.h"
#include "common.h"

namespace gpmi {
// defined in api.hip
template <typename T>
int super_factor_block(gpmi_ctx* c, T* blk, int64_t ld, int64_t w, T* linv, T* invdiag, T* lw, int64_t pivot_base);

namespace {

template <typename T>
__global__ __launch_bounds__(256) void row_sumsq_acc_kernel(const T* __restrict__ R, int64_t ldr, int64_t n, double* __restrict__ acc) {
    __shared__ double sh[256];
    const T* r = R + (int64_t)blockIdx.x * ldr;
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) {
        const double v = (double)r[j];
        s += v * v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[blockIdx.x] += sh[0];
}

template <typename T>
__global__ __launch_bounds__(1024) void dot_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n, double* __restrict__ out) {
    __shared__ double sh[1024];
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 1024) s += (double)a[j] * (double)b[j];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

template <typename T>
__global__ void identity_rows_kernel(T* __restrict__ R, int64_t ldr, int64_t nrows, int64_t col_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) R[i * ldr + col_off + i] = T(1);
}

// Wt <- w (a_i a_j - sgn Wt) in place; diagonal blocks add sum_{i < ntrace} (a_i^2 - K^-1_ii) to *trace_acc (one workgroup
// does the trace, in a fixed order: deterministic)
template <typename T>
__global__ __launch_bounds__(256) void qblock_kernel(T* __restrict__ Wt, int64_t ld, int64_t rows, int64_t cols, const T* __restrict__ ar,
                                                     const T* __restrict__ ac, double w, double sgn) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows && j < cols) {
        const double q = (double)ar[i] * (double)ac[j] - sgn * (double)Wt[i * ld + j];
        Wt[i * ld + j] = (T)(w * q);
    }
}
template <typename T>
__global__ __launch_bounds__(1024) void qtrace_kernel(const T* __restrict__ Wt, int64_t ld, int64_t ntrace, const T* __restrict__ ar, double sgn,
                                                      double* __restrict__ trace_acc) {
    __shared__ double sh[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < ntrace; i += 1024) s += (double)ar[i] * (double)ar[i] - sgn * (double)Wt[i * ld + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) trace_acc[0] += sh[0];
}
// host scalar all-reduces carry at most this many doubles (the gradient sends its n_hyp + 1 sums in chunks of 64, blocked.cpp)
constexpr int HOST_RED_CAP = 256;
__global__ void acc_add_kernel(double* __restrict__ out, const double* __restrict__ add, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] += add[i];
}

template <typename T>
struct HipDev : Dev {
    gpmi_ctx* c;
    hipStream_t main_s = nullptr;
    double* partial = nullptr;   // dmll partials
    int64_t partial_cap = 0;
    explicit HipDev(gpmi_ctx* ctx) : c(ctx) { es = (int)sizeof(T); }
    ~HipDev() override {
        if (partial) hipFree(partial);
    }
    ProfScope* open_phase[GPMI_PROF_NCLASS] = {nullptr};
    void phase(int cls, bool begin) override {  // a marker-event pair on the current stream (gpmi_profile_enable(ctx, 1 | 65))
        if (cls < GPMI_PROF_STEP_U1 || cls >= GPMI_PROF_NCLASS) return;
        if (begin) {
            if (!c->prof_on || open_phase[cls]) return;
            open_phase[cls] = new ProfScope(c, cls, 0.0);
        } else if (open_phase[cls]) {
            delete open_phase[cls];
            open_phase[cls] = nullptr;
        }
    }
    void note(hipError_t e, const char* what) {
        if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e);
    }
    void* alloc(int64_t bytes) override {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, (size_t)bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            err = std::string("hipMalloc of ") + std::to_string(bytes >> 20) + " MiB: " + hipGetErrorString(e);
            return nullptr;
        }
        return p;
    }
    void release(void* p) override { (void)hipFree(p); }
    void zero(void* p, int64_t bytes) override { note(hipMemsetAsync(p, 0, (size_t)bytes, c->stream), "hipMemsetAsync"); }
    void copy2d(void* dst, int64_t dp, const void* src, int64_t sp, int64_t w, int64_t rows) override {
        if (rows <= 0 || w <= 0) return;
        if (rows == 1 || (dp == w && sp == w)) {  // contiguous: one linear copy
            note(hipMemcpyAsync(dst, src, (size_t)(w * rows), hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync D2D");
            return;
        }
        if (w >= (64 << 10) && rows <= 256) {  // a few very long rows (whole blocks of a panel): linear copies, not a 2-D blit
            for (int64_t r = 0; r < rows; ++r)
                note(hipMemcpyAsync((char*)dst + r * dp, (const char*)src + r * sp, (size_t)w, hipMemcpyDeviceToDevice, c->stream),
                     "hipMemcpyAsync D2D");
            return;
        }
        note(hipMemcpy2DAsync(dst, (size_t)dp, src, (size_t)sp, (size_t)w, (size_t)rows, hipMemcpyDeviceToDevice, c->stream), "hipMemcpy2DAsync");
    }
    void upload(void* dst, const void* host, int64_t bytes) override {
        note(hipMemcpyAsync(dst, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream), "hipMemcpyAsync H2D");
        note(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    }
    void download(void* host, const void* src, int64_t bytes) override {
        note(hipMemcpyAsync(host, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync D2H");
        note(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    }
    // ---- streams ----
    bool is_masked = false;
    bool solo = false;
    void whole_cus(bool on) override { is_masked = set_lookahead_mode(c, on) == 1; }
    void set_world(int world) override { solo = world == 1; }
    bool masked() const { return is_masked && c->upd_stream && c->side_masked; }
    void begin_call() override {
        (void)hipSetDevice(c->device);
        if (!main_s) main_s = c->stream;
        c->stream = main_s;
        c->num_cus = full_cus();
        c->beside_update = false;
        c->gemm_reserve = 0;
        c->update_late_wgs = false;
        la_reset(c);
    }
    int full_cus_ = 0;
    int full_cus() {
        if (!full_cus_) full_cus_ = c->num_cus;
        return full_cus_;
    }
    void use(DevStream s) override {
        c->beside_update = false;
        c->side_one_per_xcd = false;
        c->chain_wide_ok = false;
        c->gemm_reserve = 0;
        c->update_late_wgs = false;
        c->num_cus = full_cus();
        switch (s) {
            case DS_MAIN: c->stream = main_s; break;
            case DS_UPD:
                if (masked()) {
                    c->stream = c->upd_stream;
                    c->num_cus = full_cus() - c->reserved_cus;
                } else {
                    c->stream = main_s;
                    c->gemm_reserve = c->side_stream ? c->lookahead_slots : 0;
                    // one rank: the 256 x 128 update's grid covers every compute unit (update256.hip: the workgroups whose unit the chain
                    // holds start when the chain's exits; gemm() below puts chain_wait_kernel in front so that the chain is placed first)
                    c->update_late_wgs = solo && c->update_full_grid && c->gemm_reserve > 0;
                }
                break;
            case DS_SIDE:
                c->stream = masked() ? c->side_masked : (c->side_stream ? c->side_stream : main_s);
                c->beside_update = c->stream != main_s;
                // free slots beside an update that will run in 256 x 128 tiles (one workgroup per CU: the free slots are whole CUs, one
                // per XCD): no chain launch of more than one workgroup per XCD (chol.h / common.h side_slots).  The driver only stays in
                // this mode while the update is long enough for that kernel (blocked.cpp kWholeCusBelow).
                c->side_one_per_xcd = !masked() && c->update256 != 0 && c->beside_update;
                c->chain_wide_ok = solo && c->update_full_grid && c->side_one_per_xcd;
                break;
            // the exchange: the main stream is idle while a factorisation runs on UPD / SIDE, so with whole CUs reserved the
            // collectives go there (their kernels find the reserved CUs free); with free slots they follow the chain
            case DS_COMM: c->stream = masked() ? main_s : (c->side_stream ? c->side_stream : main_s); break;
        }
    }
    DevEvent record() override {
        hipEvent_t e = la_event(c);
        note(hipEventRecord(e, c->stream), "hipEventRecord");
        return (DevEvent)e;
    }
    void wait(DevEvent e) override {
        if (e) note(hipStreamWaitEvent(c->stream, (hipEvent_t)e, 0), "hipStreamWaitEvent");
    }
    void sync() override {
        for (hipStream_t s : {c->upd_stream, c->side_masked, c->side_stream, main_s})
            if (s) note(hipStreamSynchronize(s), "hipStreamSynchronize");
        note(hipGetLastError(), "device");
        c->stream = main_s;
        c->num_cus = full_cus();
        c->beside_update = false;
        c->gemm_reserve = 0;
        c->update_late_wgs = false;
    }
    void* native_stream() override { return (void*)c->stream; }
    // ---- ops ----
    int set_kernel(const gpmi_kernel* k, int d, double* kdiag, int* n_hyp) override {
        const int rc = upload_program(c, k, d);
        if (rc != GPMI_OK) {
            err = c->err;
            return rc;
        }
        *kdiag = c->h_prog->kdiag;
        *n_hyp = c->h_prog->n_hyp;
        return GPMI_OK;
    }
    void assemble(const void* x, int64_t n, int d, int64_t row_off, int64_t nrows, double nugget, const double* nvec, void* A, int64_t ld,
                  int64_t ncols) override {
        const int64_t na = std::max<int64_t>(0, std::min<int64_t>(nrows, n - row_off));
        const int64_t xoff = std::min<int64_t>(row_off, n - 1) * d;
        launch_cov<T>(c, (const T*)x + xoff, na, (const T*)x, n, d, (T*)A, ld, nrows, ncols, COV_LOWER | COV_NUGGET | COV_PAD_IDENTITY, nugget, nvec,
                      row_off);
    }
    void cov_rows(const void* xa, int64_t na, const void* xb, int64_t nb, int d, void* C, int64_t ldc, int64_t ncols_total) override {
        launch_cov<T>(c, (const T*)xa, na, (const T*)xb, nb, d, (T*)C, ldc, na, ncols_total, 0, 0.0, nullptr);
    }
    int super_factor(void* blk, int64_t ld, int64_t w, void* linv, void* invd, void* lw, int64_t pivot_base) override {
        const int rc = super_factor_block<T>(c, (T*)blk, ld, w, (T*)linv, (T*)invd, (T*)lw, pivot_base);
        if (rc != GPMI_OK && err.empty()) err = c->err;
        return rc;
    }
    void gemm(void* C, int64_t ldc, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, DevShape s,
              int flags) override {
        if (M <= 0 || N <= 0 || K <= 0) return;
        // free-slot mode: the update that a chain launch is to run beside waits until the chain's workgroups are placed (chain.hip)
        if (c->chain_wait_pending && c->stream == main_s && !c->beside_update) launch_chain_wait(c);
        TileShape sh{0, 0, s.mode, s.g0, s.G, s.nstair, s.tpb > 0 ? s.tpb : 2};
        launch_gemm_shape<T>(c, (T*)C, ldc, (const T*)A, lda, (const T*)B, ldb, M, N, K, sh, c->d_info, flags);
    }
    void bsolve_block(const void* Lrows, int64_t ld, int64_t c0, int64_t nb, const void* linv, const void* lw, void* z, void* alpha) override {
        // 256 columns per launch through the diagonal 256 x 256 blocks of the block's explicit inverse (they ARE the inverses of
        // the factor's diagonal 256-blocks): N / 256 launches per solve instead of N / 64 through the 64 x 64 inverses
        (void)linv;
        for (int64_t j = nb - NB; j >= 0; j -= NB)
            launch_bsolve256<T>(c, (const T*)Lrows + j * ld, ld, c0 + j, (int)NB, (const T*)lw + j * nb + j, (T*)z, (T*)alpha, nb);
    }
    double logdiag_sum(const void* A, int64_t ld, int64_t nrows, int64_t col_off) override {
        launch_logdiag<T>(c, (const T*)A, ld, nrows, col_off, c->d_scal);
        note(hipMemcpyAsync(c->h_scal, c->d_scal, sizeof(double), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
        note(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        return c->h_scal[0];
    }
    int64_t info(bool reset) override {
        if (reset) {
            note(hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream), "hipMemsetAsync");
            return 0;
        }
        int h = 0;
        note(hipMemcpyAsync(&h, c->d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
        note(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        if (h < 0) {  // chain.hip: a dependency wait ran into its bound — a device error, not a pivot
            if (err.empty()) err = "chain kernel: a dependency wait timed out (GPMI_CHAIN=0 selects the multi-launch chain)";
            return 0;
        }
        return h;
    }
    void row_gemv(const void* R, int64_t ldr, int64_t P, int64_t n, const void* v, const void* add, void* out) override {
        launch_row_gemv<T>(c, (const T*)R, ldr, P, n, (const T*)v, (const T*)add, (T*)out);
    }
    void row_sumsq_acc(const void* R, int64_t ldr, int64_t P, int64_t n, double* acc) override {
        if (P > 0) hipLaunchKernelGGL(row_sumsq_acc_kernel<T>, dim3((unsigned)P), dim3(256), 0, c->stream, (const T*)R, ldr, n, acc);
    }
    double dot(const void* a, const void* b, int64_t n) override {
        hipLaunchKernelGGL(dot_kernel<T>, dim3(1), dim3(1024), 0, c->stream, (const T*)a, (const T*)b, n, c->d_scal + 4);
        note(hipMemcpyAsync(c->h_scal + 4, c->d_scal + 4, sizeof(double), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync");
        note(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        return c->h_scal[4];
    }
    void set_identity_rows(void* R, int64_t ldr, int64_t nrows, int64_t col_off) override {
        note(hipMemsetAsync(R, 0, (size_t)(nrows * ldr) * sizeof(T), c->stream), "hipMemsetAsync");
        hipLaunchKernelGGL(identity_rows_kernel<T>, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream, (T*)R, ldr, nrows, col_off);
    }
    void qblock(void* Wt, int64_t ld, int64_t rows, int64_t cols, const void* ar, const void* ac, double w, bool neg, bool diag, int64_t ntrace,
                double* trace_acc) override {
        if (rows <= 0 || cols <= 0) return;
        const double sgn = neg ? -1.0 : 1.0;
        if (diag && trace_acc && ntrace > 0)
            hipLaunchKernelGGL(qtrace_kernel<T>, dim3(1), dim3(1024), 0, c->stream, (const T*)Wt, ld, ntrace, (const T*)ar, sgn, trace_acc);
        hipLaunchKernelGGL(qblock_kernel<T>, dim3((unsigned)((cols + 255) / 256), (unsigned)rows), dim3(256), 0, c->stream, (T*)Wt, ld, rows, cols,
                           (const T*)ar, (const T*)ac, w, sgn);
    }
    void dmll_rect_acc(const void* xa, int64_t na, const void* xb, int64_t nb, int d, const void* Wt, int64_t ld, int n_hyp, double* out) override {
        if (na <= 0 || nb <= 0) return;
        const int64_t nblocks = ((na + 63) / 64) * ((nb + 63) / 64);
        const int64_t need = (nblocks + 1) * (int64_t)(n_hyp + 1) * 8;
        if (partial_cap < need) {
            note(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
            if (partial) (void)hipFree(partial);
            partial = nullptr;
            partial_cap = 0;
            if (hipMalloc(&partial, (size_t)(2 * need)) != hipSuccess) {
                (void)hipGetLastError();
                if (err.empty()) err = "out of device memory (gradient partial sums)";
                return;
            }
            partial_cap = 2 * need;
        }
        const int64_t nb2 = launch_dmll_rect<T>(c, (const T*)xa, na, (const T*)xb, nb, d, (const T*)Wt, ld, partial, n_hyp);
        double* red = partial + nb2 * (n_hyp + 1);
        launch_reduce_partials(c, partial, nb2, n_hyp + 1, red);
        hipLaunchKernelGGL(acc_add_kernel, dim3(1), dim3(256), 0, c->stream, out, (const double*)red, n_hyp);
    }
};

// ---- RCCL, opened at run time (no link-time dependency: a single-GPU user never loads it) --------------------------------
struct Rccl {
    typedef struct { char internal[128]; } UniqueId;
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy a host program (torch) already loaded, if any
            if (lib) break;
        }
        if (!lib)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
            }
        if (!lib) {
            err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");  // optional: without them the gathers of a step go out one by one
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !Broadcast || !AllGather || !AllReduce) {
            err = "librccl.so lacks an nccl* entry point";
            return false;
        }
        return true;
    }
};
static Rccl g_rccl;

// nccl data types / ops by value (nccl.h): ncclInt8 0 / ncclChar 0, ncclFloat32 7, ncclFloat64 8; ncclSum 0, ncclMax 2, ncclMin 3
struct RcclComm : Comm {
    void* comm = nullptr;
    gpmi_ctx* ctx;
    double* d_tmp = nullptr;
    explicit RcclComm(gpmi_ctx* c) : ctx(c) {}
    ~RcclComm() override {
        if (comm) g_rccl.CommDestroy(comm);
        if (d_tmp) (void)hipFree(d_tmp);
    }
    int broadcast(void* buf, int64_t bytes, int root, void* stream) override {
        return g_rccl.Broadcast(buf, buf, (size_t)bytes, 0, root, comm, (hipStream_t)stream);
    }
    int all_gather(const void* send, void* recv, int64_t bytes_each, void* stream) override {
        return g_rccl.AllGather(send, recv, (size_t)bytes_each, 0, comm, (hipStream_t)stream);
    }
    void group_begin() override {
        if (g_rccl.GroupStart && g_rccl.GroupEnd) (void)g_rccl.GroupStart();
    }
    int group_end() override { return (g_rccl.GroupStart && g_rccl.GroupEnd) ? g_rccl.GroupEnd() : 0; }
    int all_reduce_sum(void* buf, int64_t count, int es, void* stream) override {
        return g_rccl.AllReduce(buf, buf, (size_t)count, es == 8 ? 8 : 7, 0, comm, (hipStream_t)stream);
    }
    int host_allreduce(double* vals, int n, int op) override {
        if (n > HOST_RED_CAP) return 1;
        if (!d_tmp && hipMalloc(&d_tmp, HOST_RED_CAP * sizeof(double)) != hipSuccess) return 1;
        hipStream_t s = ctx->stream;
        if (hipMemcpyAsync(d_tmp, vals, (size_t)n * 8, hipMemcpyHostToDevice, s) != hipSuccess) return 1;
        const int rc = g_rccl.AllReduce(d_tmp, d_tmp, (size_t)n, 8, op == 0 ? 0 : (op == 1 ? 3 : 2), comm, s);
        if (rc) return rc;
        if (hipMemcpyAsync(vals, d_tmp, (size_t)n * 8, hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
        return hipStreamSynchronize(s) == hipSuccess ? 0 : 1;
    }
};

}  // namespace

// what a blocked gpmi_gp owns (common.h: gpmi_gp::blocked)
struct BlockedHandle {
    std::unique_ptr<Dev> dev;
    std::unique_ptr<BlockedGP> gp;
};
void blocked_destroy(void* p) { delete (BlockedHandle*)p; }

// ---- in-process device group -------------------------------------------------------------------------------------------
// gpmi_ctx_create(n_devices > 1): SURVEY 8(b)'s single-process multi-GPU form.  One context per device id (the same id may
// repeat: that is how the single-GPU test box runs it), one BlockedGP per member, one worker thread per member for the
// duration of an API call, and a communicator that needs no library: collectives are PEER COPIES ordered by events — a rank
// publishes its buffer and a "ready" event, the readers make their stream wait for it and copy straight from the owner's
// memory (hipMemcpyAsync device-to-device: one xGMI hop per pair, all pairs at once), then publish "done" events the owner's
// stream waits for before it may touch the buffer again.  The host threads only rendezvous to exchange the handles.
struct LocalGroup {
    int n = 0;
    std::vector<gpmi_ctx*> members;  // [0] = the primary
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t gen = 0;
    bool aborted = false;
    // slots published by the ranks, two sets used alternately (collective parity) so that a rank still reading the slots of
    // collective i cannot race a fast rank publishing collective i + 1
    std::vector<const void*> ptr[2];
    std::vector<hipEvent_t> ready[2], done[2];
    std::vector<double> hv[2];
    // test hook (see LocalComm::inject_delay): read ONCE per API call by the calling thread (group_run), never by the worker threads
    long long delay_us = 0;
    int delay_on = 3;
    bool barrier() {  // false: the group was aborted (a member failed outside a collective)
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t g0 = gen;
        if (++waiting == n) {
            waiting = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        cv.wait(lk, [&] { return gen != g0 || aborted; });
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
    void reset() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = false;
        waiting = 0;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void sum_ranks_kernel(T* __restrict__ out, const T* __restrict__ parts, int64_t count, int world) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    T s = parts[i];
    for (int q = 1; q < world; ++q) s += parts[(int64_t)q * count + i];  // fixed order: bit-identical on every rank
    out[i] = s;
}

// TEST HOOK (tests/test_gpu_dist.py: injected-latency overlap test): a kernel that occupies one wave for `us` microseconds of the
// constant 100 MHz wall clock, enqueued in front of a collective on the stream the collective was given
__global__ void comm_delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

struct LocalComm : Comm {
    LocalGroup* g;
    gpmi_ctx* c;
    std::vector<hipEvent_t> pool;
    size_t next = 0;
    uint64_t seq = 0;  // collectives issued so far (parity selects the slot set; every rank issues the same sequence)
    void* red = nullptr;
    int64_t red_cap = 0;
    LocalComm(LocalGroup* grp, gpmi_ctx* ctx, int r) : g(grp), c(ctx) {
        rank = r;
        world = grp->n;
    }
    ~LocalComm() override {
        (void)hipSetDevice(c->device);
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
        if (red) (void)hipFree(red);
    }
    hipEvent_t event() {
        if (pool.size() < 64) {
            hipEvent_t e;
            (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
            pool.push_back(e);
            return e;
        }
        return pool[next++ % pool.size()];
    }
    // every rank reads `bytes` from each published pointer in `from` (rank index -> destination), then the owners wait for the readers
    int exchange(const void* mine, void* stream, int64_t bytes, int root /* -1: all-gather */, void* dst_base) {
        hipStream_t s = (hipStream_t)stream;
        const int par = (int)(seq++ & 1);
        hipEvent_t rdy = event();
        if (hipEventRecord(rdy, s) != hipSuccess) return 1;
        g->ptr[par][rank] = mine;
        g->ready[par][rank] = rdy;
        if (!g->barrier()) return 2;
        for (int q = 0; q < world; ++q) {
            if (root >= 0 && q != root) continue;
            char* dst = (char*)dst_base + (root >= 0 ? 0 : (int64_t)q * bytes);
            if (q == rank) {
                if (root < 0 && dst != mine && hipMemcpyAsync(dst, mine, (size_t)bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
                continue;
            }
            if (hipStreamWaitEvent(s, g->ready[par][q], 0) != hipSuccess) return 1;
            if (hipMemcpyAsync(dst, g->ptr[par][q], (size_t)bytes, hipMemcpyDefault, s) != hipSuccess) return 1;
        }
        hipEvent_t dn = event();
        if (hipEventRecord(dn, s) != hipSuccess) return 1;
        g->done[par][rank] = dn;
        if (!g->barrier()) return 2;
        for (int q = 0; q < world; ++q)  // my buffer may change only after everyone who reads it has
            if (q != rank && (root < 0 || rank == root))
                if (hipStreamWaitEvent(s, g->done[par][q], 0) != hipSuccess) return 1;
        return 0;
    }
    // GPMI_TEST_COMM_DELAY_US / _ON (read once per API call by the calling thread — group_run —, so a test can change them between
    // fits and no worker thread ever calls getenv): extra latency in front of every inverse broadcast and every panel exchange (its
    // per-group gathers are one exchange: the delay goes in front of the first) — how tests measure what the look-ahead pipeline of
    // blocked.cpp really hides.  Only the in-process communicator of a device group has the hook.
    long long delayed = 0;
    bool in_group = false, group_delayed = false;
    void inject_delay(void* stream, int what /* 1 broadcast, 2 panel exchange */) {
        const long long us = g->delay_us;
        if (us <= 0 || !(g->delay_on & what)) return;
        hipLaunchKernelGGL(comm_delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, us * 100);
        ++delayed;
    }
    void group_begin() override {
        in_group = true;
        group_delayed = false;
    }
    int group_end() override {
        in_group = false;
        return 0;
    }
    int broadcast(void* buf, int64_t bytes, int root, void* stream) override {
        inject_delay(stream, 1);
        return exchange(buf, stream, bytes, root, buf);
    }
    int all_gather(const void* send, void* recv, int64_t bytes_each, void* stream) override {
        if (!in_group || !group_delayed) inject_delay(stream, 2);
        group_delayed = true;
        return exchange(send, stream, bytes_each, -1, recv);
    }
    int all_reduce_sum(void* buf, int64_t count, int es, void* stream) override {
        const int64_t need = (int64_t)world * count * es;
        if (red_cap < need) {
            (void)hipStreamSynchronize((hipStream_t)stream);
            if (red) (void)hipFree(red);
            red = nullptr;
            red_cap = 0;
            if (hipMalloc(&red, (size_t)need) != hipSuccess) return 1;
            red_cap = need;
        }
        const int rc = exchange(buf, stream, count * es, -1, red);
        if (rc) return rc;
        const unsigned blocks = (unsigned)((count + 255) / 256);
        if (es == 8)
            hipLaunchKernelGGL(sum_ranks_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double*)buf, (const double*)red, count, world);
        else
            hipLaunchKernelGGL(sum_ranks_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)buf, (const float*)red, count, world);
        return 0;
    }
    int host_allreduce(double* vals, int n, int op) override {
        if (n > HOST_RED_CAP) return 1;
        const int par = (int)(seq++ & 1);
        for (int i = 0; i < n; ++i) g->hv[par][(size_t)rank * HOST_RED_CAP + i] = vals[i];
        if (!g->barrier()) return 2;
        for (int i = 0; i < n; ++i) {
            double a = g->hv[par][i];
            for (int q = 1; q < world; ++q) {
                const double b = g->hv[par][(size_t)q * HOST_RED_CAP + i];
                a = op == 0 ? a + b : (op == 1 ? (b < a ? b : a) : (b > a ? b : a));
            }
            vals[i] = a;
        }
        return g->barrier() ? 0 : 2;
    }
};

struct GroupHandle {
    LocalGroup* g = nullptr;
    std::vector<std::unique_ptr<LocalComm>> comms;
    std::vector<std::unique_ptr<BlockedHandle>> ranks;
};

BlockedGP* blocked_of(gpmi_gp* gp) {
    if (gp && gp->group) return ((GroupHandle*)gp->group)->ranks[0]->gp.get();
    return gp && gp->blocked ? ((BlockedHandle*)gp->blocked)->gp.get() : nullptr;
}

int group_create(gpmi_ctx* primary, int n, const int* ids) {
    LocalGroup* g = new LocalGroup();
    g->n = n;
    g->members.assign((size_t)n, nullptr);
    g->members[0] = primary;
    primary->group = g;
    primary->group_rank = 0;
    for (int r = 1; r < n; ++r) {
        gpmi_ctx* m = nullptr;
        const int rc = create_member_context(ids[r], &m);
        if (rc != GPMI_OK) {
            primary->err = "gpmi_ctx_create: could not create the context of device " + std::to_string(ids[r]);
            return rc;  // the caller destroys the primary, which takes the members made so far with it
        }
        m->group = g;
        m->group_rank = r;
        g->members[(size_t)r] = m;
    }
    for (int par = 0; par < 2; ++par) {
        g->ptr[par].assign((size_t)n, nullptr);
        g->ready[par].assign((size_t)n, nullptr);
        g->done[par].assign((size_t)n, nullptr);
        g->hv[par].assign((size_t)n * HOST_RED_CAP, 0.0);
    }
    // direct peer copies between distinct devices (errors are not fatal: the runtime then stages through the host)
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            if ((ids[a] & 255) != (ids[b] & 255)) {  // (ids may carry a CU-partition code above bit 8)
                (void)hipSetDevice(ids[a] & 255);
                if (hipDeviceEnablePeerAccess(ids[b] & 255, 0) != hipSuccess) (void)hipGetLastError();
            }
    (void)hipSetDevice(primary->device);
    return GPMI_OK;
}

std::vector<gpmi_ctx*> group_members(gpmi_ctx* c) {
    if (c->group) return ((LocalGroup*)c->group)->members;
    return {c};
}

void group_destroy(gpmi_ctx* primary) {
    LocalGroup* g = (LocalGroup*)primary->group;
    if (!g) return;
    for (size_t r = 1; r < g->members.size(); ++r)
        if (g->members[r]) {
            g->members[r]->group = nullptr;
            gpmi_ctx_destroy(g->members[r]);
        }
    primary->group = nullptr;
    delete g;
}

// run f(rank) on one thread per member (rank 0 on the caller's); a member that fails outside a collective aborts the group's
// rendezvous so that nobody waits for it.  Returns rank 0's status unless another member failed harder.
template <typename F>
static int group_run(GroupHandle* h, F f) {
    LocalGroup* g = h->g;
    g->reset();
    {   // the injected-latency test hook, sampled here on the caller's thread (not per collective, not on worker threads)
        const char* e = getenv("GPMI_TEST_COMM_DELAY_US");
        g->delay_us = e ? atoll(e) : 0;
        const char* on = g->delay_us > 0 ? getenv("GPMI_TEST_COMM_DELAY_ON") : nullptr;  // 1: broadcasts only, 2: panel exchanges only, default both
        g->delay_on = on ? atoi(on) : 3;
    }
    // a member that left the previous call early (EDEVICE / EARG -> abort) has issued fewer collectives than the others: every
    // call starts from the same slot parity again (all streams were drained when the previous call returned)
    for (auto& cm : h->comms) cm->seq = 0;
    std::vector<int> rc((size_t)g->n, GPMI_OK);
    auto body = [&](int r) {
        (void)hipSetDevice(g->members[(size_t)r]->device);
        rc[(size_t)r] = f(r);
        if (rc[(size_t)r] == GPMI_EDEVICE || rc[(size_t)r] == GPMI_EARG) g->abort();
    };
    std::vector<std::thread> th;
    for (int r = 1; r < g->n; ++r) th.emplace_back(body, r);
    body(0);
    for (auto& t : th) t.join();
    (void)hipSetDevice(g->members[0]->device);
    int out = rc[0];
    for (int r = 1; r < g->n; ++r)
        if (rc[(size_t)r] > out || (out == GPMI_OK && rc[(size_t)r] != GPMI_OK)) {
            out = rc[(size_t)r];
            g->members[0]->err = h->ranks[(size_t)r]->gp->error();
        }
    if (rc[0] != GPMI_OK && out == rc[0]) g->members[0]->err = h->ranks[0]->gp->error();
    return out;
}

int group_gp_create(gpmi_ctx* primary, int dtype, int d, int64_t n, const void* x, int64_t block_rows, int stripe_blocks, gpmi_gp** out) {
    LocalGroup* g = (LocalGroup*)primary->group;
    std::unique_ptr<GroupHandle> h(new GroupHandle());
    h->g = g;
    BlockedOpts o;
    o.block = block_rows;
    o.stripe_blocks = stripe_blocks;
    for (int r = 0; r < g->n; ++r) {
        gpmi_ctx* m = g->members[(size_t)r];
        h->comms.emplace_back(new LocalComm(g, m, r));
        std::unique_ptr<BlockedHandle> b(new BlockedHandle());
        if (dtype == 64)
            b->dev.reset(new HipDev<double>(m));
        else
            b->dev.reset(new HipDev<float>(m));
        b->gp.reset(new BlockedGP(b->dev.get(), h->comms.back().get(), d, n, o));
        h->ranks.push_back(std::move(b));
    }
    const int rc = group_run(h.get(), [&](int r) { return h->ranks[(size_t)r]->gp->init(x); });
    if (rc != GPMI_OK) return rc;
    gpmi_gp* gp = new gpmi_gp();
    gp->ctx = primary;
    gp->dtype = dtype;
    gp->d = d;
    gp->n = n;
    gp->group = h.release();
    *out = gp;
    return GPMI_OK;
}

void group_gp_destroy(gpmi_gp* gp) {
    GroupHandle* h = (GroupHandle*)gp->group;
    if (!h) return;
    for (size_t r = 0; r < h->ranks.size(); ++r) {
        (void)hipSetDevice(h->g->members[r]->device);
        h->ranks[r].reset();
    }
    h->comms.clear();
    (void)hipSetDevice(h->g->members[0]->device);
    delete h;
    gp->group = nullptr;
}

int group_fit(gpmi_gp* gp, const gpmi_kernel* k, const double* log_noise, int64_t n_noise, const void* ymu, double* mll_out, void* alpha_out,
              int64_t* info_out) {
    GroupHandle* h = (GroupHandle*)gp->group;
    std::vector<int64_t> info((size_t)h->g->n, 0);
    std::vector<double> mll((size_t)h->g->n, 0.0);
    const int rc = group_run(h, [&](int r) {
        return h->ranks[(size_t)r]->gp->fit(k, log_noise, n_noise, ymu, &mll[(size_t)r], r == 0 ? alpha_out : nullptr, &info[(size_t)r]);
    });
    if (info_out) *info_out = info[0];
    if (rc == GPMI_OK && mll_out) *mll_out = mll[0];
    return rc;
}
int group_predict(gpmi_gp* gp, const gpmi_kernel* k, int64_t p, const void* xpred, const void* mean_pred, int full_cov, void* mu_out, void* var_out) {
    GroupHandle* h = (GroupHandle*)gp->group;
    const size_t es = gp->dtype == 64 ? 8 : 4;
    // every rank produces the (replicated) results; the members other than rank 0 write into scratch
    std::vector<std::vector<char>> mu((size_t)h->g->n), var((size_t)h->g->n);
    for (int r = 1; r < h->g->n; ++r) {
        mu[(size_t)r].resize((size_t)p * es);
        var[(size_t)r].resize((size_t)(full_cov ? p * p : p) * es);
    }
    return group_run(h, [&](int r) {
        return h->ranks[(size_t)r]->gp->predict(k, p, xpred, mean_pred, full_cov, r == 0 ? mu_out : (void*)mu[(size_t)r].data(),
                                                r == 0 ? var_out : (void*)var[(size_t)r].data());
    });
}
int group_grad(gpmi_gp* gp, const gpmi_kernel* k, const double* log_noise, int64_t n_noise, double* dkern_out, int n_kern, double* dnoise_out) {
    GroupHandle* h = (GroupHandle*)gp->group;
    std::vector<std::vector<double>> dk((size_t)h->g->n, std::vector<double>((size_t)std::max(n_kern, 1)));
    std::vector<double> dn((size_t)h->g->n, 0.0);
    const int rc = group_run(h, [&](int r) {
        return h->ranks[(size_t)r]->gp->grad(k, log_noise, n_noise, r == 0 ? dkern_out : dk[(size_t)r].data(), n_kern,
                                             dnoise_out ? (r == 0 ? dnoise_out : &dn[(size_t)r]) : nullptr);
    });
    return rc;
}
int group_factor_diag(gpmi_gp* gp, void* out) {
    GroupHandle* h = (GroupHandle*)gp->group;
    const size_t es = gp->dtype == 64 ? 8 : 4;
    std::vector<std::vector<char>> tmp((size_t)h->g->n);
    for (int r = 1; r < h->g->n; ++r) tmp[(size_t)r].resize((size_t)gp->n * es);
    return group_run(h, [&](int r) { return h->ranks[(size_t)r]->gp->factor_diag(r == 0 ? out : (void*)tmp[(size_t)r].data()); });
}
// the AbstractPDMat surface: every member receives the same right-hand sides and produces the (replicated) result; the members other than
// rank 0 work on copies
int group_solve(gpmi_gp* gp, int64_t nrhs, void* b, bool backward) {
    GroupHandle* h = (GroupHandle*)gp->group;
    const size_t bytes = (size_t)gp->n * (size_t)nrhs * (gp->dtype == 64 ? 8 : 4);
    std::vector<std::vector<char>> tmp((size_t)h->g->n);
    for (int r = 1; r < h->g->n; ++r) tmp[(size_t)r].assign((const char*)b, (const char*)b + bytes);
    return group_run(h, [&](int r) { return h->ranks[(size_t)r]->gp->solve(nrhs, r == 0 ? b : (void*)tmp[(size_t)r].data(), backward); });
}
int group_update_alpha(gpmi_gp* gp, const void* ymu, double* mll_out, void* alpha_out) {
    GroupHandle* h = (GroupHandle*)gp->group;
    std::vector<double> mll((size_t)h->g->n, 0.0);
    const int rc = group_run(h, [&](int r) { return h->ranks[(size_t)r]->gp->update_alpha(ymu, &mll[(size_t)r], r == 0 ? alpha_out : nullptr); });
    if (mll_out) *mll_out = mll[0];
    return rc;
}
int group_inv_diag(gpmi_gp* gp, void* out) {
    GroupHandle* h = (GroupHandle*)gp->group;
    const size_t es = gp->dtype == 64 ? 8 : 4;
    std::vector<std::vector<char>> tmp((size_t)h->g->n);
    for (int r = 1; r < h->g->n; ++r) tmp[(size_t)r].resize((size_t)gp->n * es);
    return group_run(h, [&](int r) { return h->ranks[(size_t)r]->gp->inv_diag(r == 0 ? out : (void*)tmp[(size_t)r].data()); });
}
int group_factor_to_host(gpmi_gp* gp, void* U_out) {
    GroupHandle* h = (GroupHandle*)gp->group;
    const size_t es = gp->dtype == 64 ? 8 : 4;
    // ONE scratch image shared by the members other than rank 0 would race; each gets its own (n x n on the host: this entry point is for
    // inspection at sizes where that is affordable — the factor of a large model stays on the devices)
    std::vector<std::vector<char>> tmp((size_t)h->g->n);
    for (int r = 1; r < h->g->n; ++r) tmp[(size_t)r].resize((size_t)gp->n * (size_t)gp->n * es);
    return group_run(h, [&](int r) { return h->ranks[(size_t)r]->gp->factor_to_host(r == 0 ? U_out : (void*)tmp[(size_t)r].data()); });
}

}  // namespace gpmi

struct gpmi_comm {
    std::unique_ptr<gpmi::Comm> impl;
};

using namespace gpmi;

extern "C" {

int gpmi_comm_create_callbacks(const gpmi_comm_callbacks* cb, int rank, int world, gpmi_comm** out) {
    if (!cb || !out || world < 1 || rank < 0 || rank >= world || !cb->broadcast || !cb->all_gather || !cb->all_reduce_sum || !cb->host_allreduce)
        return GPMI_EARG;
    gpmi_comm* cm = new gpmi_comm();
    cm->impl.reset(new CallbackComm(*cb, rank, world));
    *out = cm;
    return GPMI_OK;
}

int gpmi_comm_unique_id(void* id128_out) {
    if (!id128_out) return GPMI_EARG;
    if (!g_rccl.load()) return GPMI_EDEVICE;
    Rccl::UniqueId id;
    if (g_rccl.GetUniqueId(&id) != 0) return GPMI_EDEVICE;
    memcpy(id128_out, id.internal, 128);
    return GPMI_OK;
}

int gpmi_comm_create_rccl(gpmi_ctx* c, const void* id128, int rank, int world, gpmi_comm** out) {
    if (!c) return GPMI_EARG;
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) {
        c->err = "gpmi_comm_create_rccl: bad argument";
        return GPMI_EARG;
    }
    if (!g_rccl.load()) {
        c->err = g_rccl.err;
        return GPMI_EDEVICE;
    }
    GPMI_HIP(c, hipSetDevice(c->device));
    Rccl::UniqueId id;
    memcpy(id.internal, id128, 128);
    std::unique_ptr<RcclComm> rc(new RcclComm(c));
    rc->rank = rank;
    rc->world = world;
    const int e = g_rccl.CommInitRank(&rc->comm, world, id, rank);
    if (e != 0) {
        c->err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error");
        rc->comm = nullptr;
        return GPMI_EDEVICE;
    }
    gpmi_comm* cm = new gpmi_comm();
    cm->impl = std::move(rc);
    *out = cm;
    return GPMI_OK;
}

void gpmi_comm_destroy(gpmi_comm* cm) { delete cm; }

// Every collective of the communicator on small device buffers with rank-dependent patterns, verified on the host: a launcher
// calls this once before its first fit so that a broken transport fails HERE, with a message, instead of inside a factorisation.
int gpmi_comm_selftest(gpmi_ctx* c, gpmi_comm* cm) {
    if (!c) return GPMI_EARG;
    if (!cm || !cm->impl) {
        c->err = "gpmi_comm_selftest: null communicator";
        return GPMI_EARG;
    }
    Comm* k = cm->impl.get();
    GPMI_HIP(c, hipSetDevice(c->device));
    const int W = k->world, r = k->rank, n = 1000;
    double *d_a = nullptr, *d_g = nullptr;
    float* d_f = nullptr;
    GPMI_HIP(c, hipMalloc(&d_a, n * sizeof(double)));
    GPMI_HIP(c, hipMalloc(&d_g, (size_t)W * n * sizeof(double)));
    GPMI_HIP(c, hipMalloc(&d_f, n * sizeof(float)));
    std::vector<double> h(n), hg((size_t)W * n);
    std::vector<float> hf(n);
    std::string bad;
    auto fill = [&](double scale) {
        for (int i = 0; i < n; ++i) {
            h[i] = scale * (r + 1) + 0.001 * i;
            hf[i] = (float)(r + 1) + 0.5f * (float)(i % 7);
        }
        (void)hipMemcpy(d_a, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
        (void)hipMemcpy(d_f, hf.data(), n * sizeof(float), hipMemcpyHostToDevice);
    };
    hipStream_t s = c->stream;
    int rc = 0;
    // broadcast from the last rank
    fill(1.0);
    rc |= k->broadcast(d_a, n * sizeof(double), W - 1, s);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), d_a, n * sizeof(double), hipMemcpyDeviceToHost);
    for (int i = 0; i < n && bad.empty(); ++i)
        if (h[i] != 1.0 * W + 0.001 * i) bad = "broadcast";
    // all-gather
    fill(2.0);
    rc |= k->all_gather(d_a, d_g, n * sizeof(double), s);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(hg.data(), d_g, (size_t)W * n * sizeof(double), hipMemcpyDeviceToHost);
    for (int q = 0; q < W && bad.empty(); ++q)
        for (int i = 0; i < n; ++i)
            if (hg[(size_t)q * n + i] != 2.0 * (q + 1) + 0.001 * i) {
                bad = "all_gather";
                break;
            }
    // all-reduce, double and float
    fill(3.0);
    rc |= k->all_reduce_sum(d_a, n, 8, s);
    rc |= k->all_reduce_sum(d_f, n, 4, s);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), d_a, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hf.data(), d_f, n * sizeof(float), hipMemcpyDeviceToHost);
    const double tri = 0.5 * W * (W + 1);
    for (int i = 0; i < n && bad.empty(); ++i) {
        if (fabs(h[i] - (3.0 * tri + 0.001 * i * W)) > 1e-9) bad = "all_reduce_sum (double)";
        if (fabsf(hf[i] - ((float)tri + 0.5f * (float)(i % 7) * (float)W)) > 1e-3f) bad = "all_reduce_sum (float)";
    }
    // host scalars
    double v[3] = {(double)(r + 1), (double)(r + 1), (double)(r + 1)};
    rc |= k->host_allreduce(v, 1, 0);
    rc |= k->host_allreduce(v + 1, 1, 1);
    rc |= k->host_allreduce(v + 2, 1, 2);
    if (bad.empty() && (v[0] != tri || v[1] != 1.0 || v[2] != (double)W)) bad = "host_allreduce";
    (void)hipFree(d_a);
    (void)hipFree(d_g);
    (void)hipFree(d_f);
    if (rc != 0 || !bad.empty()) {
        c->err = "gpmi_comm_selftest: " + (bad.empty() ? std::string("a collective returned an error") : bad + " delivered wrong data") +
                 " (rank " + std::to_string(r) + " of " + std::to_string(W) + ")";
        return GPMI_EDEVICE;
    }
    return GPMI_OK;
}

int gpmi_gp_create_blocked(gpmi_ctx* c, gpmi_comm* comm, int dtype, int d, int64_t n, const void* x, int64_t block_rows, int stripe_blocks,
                           gpmi_gp** out) {
    if (!c) return GPMI_EARG;
    if (!out || !x || (dtype != 64 && dtype != 32) || d <= 0 || n <= 0 || block_rows < 0 || stripe_blocks < 0) {
        c->err = "gpmi_gp_create_blocked: bad argument (dtype must be 64|32, 1 <= d <= 2^20, n >= 1)";
        return GPMI_EARG;
    }
    *out = nullptr;
    GPMI_HIP(c, hipSetDevice(c->device));
    if (c->group && !comm) {  // a device group: one BlockedGP per member, joined by the in-process communicator
        if (c->group_rank != 0) {
            c->err = "gpmi_gp_create_blocked: pass the group's primary context";
            return GPMI_EARG;
        }
        return group_gp_create(c, dtype, d, n, x, block_rows, stripe_blocks, out);
    }
    std::unique_ptr<BlockedHandle> h(new BlockedHandle());
    if (dtype == 64)
        h->dev.reset(new HipDev<double>(c));
    else
        h->dev.reset(new HipDev<float>(c));
    BlockedOpts o;
    o.block = block_rows;
    o.stripe_blocks = stripe_blocks;
    h->gp.reset(new BlockedGP(h->dev.get(), comm ? comm->impl.get() : nullptr, d, n, o));
    const int rc = h->gp->init(x);
    if (rc != GPMI_OK) {
        c->err = h->gp->error();
        return rc;
    }
    gpmi_gp* gp = new gpmi_gp();
    gp->ctx = c;
    gp->dtype = dtype;
    gp->d = d;
    gp->n = n;
    gp->blocked = h.release();
    *out = gp;
    return GPMI_OK;
}

int gpmi_gp_blocked_info(gpmi_gp* gp, int64_t* block_rows, int32_t* n_stripes, int64_t* factor_bytes) {
    BlockedGP* b = blocked_of(gp);
    if (!b) {
        if (gp && gp->ctx) gp->ctx->err = "gpmi_gp_blocked_info: not a blocked handle";
        return GPMI_EARG;
    }
    if (block_rows) *block_rows = b->block_rows();
    if (n_stripes) *n_stripes = b->nstripes();
    if (factor_bytes) *factor_bytes = b->stored_bytes();
    return GPMI_OK;
}

}  // extern "C"
