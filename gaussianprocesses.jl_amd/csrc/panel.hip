// panel.hip — the latency-bound pieces around the MFMA update:
//   potf2      64 x 64 diagonal-block Cholesky, LDS/register resident, ONE wavefront
//              (the POTF2 of the blocked dpotrf behind make_posdef!, src/GP.jl:110)
//   trsm_rows  X <- X * L11^-T, one row per lane (panel solve; also whiten!, src/GP.jl:27)
//   bsolve     backward substitution L' alpha = z  (second half of cK \ y, src/GPE.jl:208)
//   finalize   logdet (PDMats: 2 sum log U_ii) + y'alpha + mll (src/GPE.jl:210)
//   row_gemv / row_var   predictive mean / variance reductions (src/GP.jl:26,75)
#include "common.h"

namespace gpmi {

namespace {

template <typename T>
__device__ __forceinline__ T tsqrt(T x);
template <>
__device__ __forceinline__ double tsqrt<double>(double x) { return sqrt(x); }
template <>
__device__ __forceinline__ float tsqrt<float>(float x) { return sqrtf(x); }

// ---------------------------------------------------------------------------------------------
// potf2: ONE wavefront; lane i owns row i of the 64 x 64 block in registers.  Step j:
//   pivot d = a_jj by v_readlane (j is a compile-time constant after unrolling);
//   1/sqrt(d) from v_rsq_f64 + two Newton steps, then sqrt(d) = d * rsqrt(d) and 1/sqrt(d) each
//   polished by one fused correction (no fp64 divide / sqrt library sequences on the critical path);
//   column j is scaled in place (dpotf2 does the same dscal by the reciprocal);
//   every other lane's l_cj is fetched with v_readlane into SGPRs and applied as
//   a_ic -= l_ij * l_cj (c > j) — no LDS round trip, no barrier, one SGPR operand per v_fma_f64.
// The reciprocals 1 / L_jj are kept in `invdiag` for the solves that follow (trsm_rows, bsolve).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T rsqrt_seed(T x);
template <>
__device__ __forceinline__ double rsqrt_seed<double>(double x) { return __builtin_amdgcn_rsq(x); }
template <>
__device__ __forceinline__ float rsqrt_seed<float>(float x) { return __builtin_amdgcn_rsqf(x); }

template <typename T>
__device__ __forceinline__ T bcast_lane(T v, int srclane);
template <>
__device__ __forceinline__ double bcast_lane<double>(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ float bcast_lane<float>(float v, int srclane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}

template <typename T>
__global__ __launch_bounds__(64) void potf2_kernel(T* __restrict__ A, int64_t ld, T* __restrict__ invdiag,
                                                   int* __restrict__ info, int64_t pivot_base) {
    if (*info != 0) return;
    __shared__ T S[64 * 65];
    const int i = threadIdx.x;
    for (int r = 0; r < 64; ++r) S[r * 65 + i] = A[(int64_t)r * ld + i];  // coalesced rows
    __syncthreads();
    T a[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) a[c] = S[i * 65 + c];

    int fail = 0;
    T myinv = T(0);
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const T d = bcast_lane<T>(a[j], j);
        if (!(d > T(0))) {  // also catches NaN; uniform across the wave
            fail = j + 1;
            break;
        }
        T r = rsqrt_seed<T>(d);
        r = r * (T(1.5) - T(0.5) * d * r * r);
        r = r * (T(1.5) - T(0.5) * d * r * r);
        T sq = d * r;
        sq = sq + (T(0.5) * r) * (d - sq * sq);       // sqrt(d)
        const T inv = r + r * (T(1) - sq * r);        // 1 / sqrt(d)
        const T lij = (i == j) ? sq : a[j] * inv;
        a[j] = lij;
        if (i == j) myinv = inv;
#pragma unroll
        for (int c = j + 1; c < 64; ++c) a[c] -= lij * bcast_lane<T>(lij, c);
    }
    if (fail) {
        if (i == 0) *info = (int)(pivot_base + fail);
        return;
    }
    invdiag[i] = myinv;
    // write back: lower triangle = L, strict upper = 0
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 64; ++c) S[i * 65 + c] = (c <= i) ? a[c] : T(0);
    __syncthreads();
    for (int r = 0; r < 64; ++r) A[(int64_t)r * ld + i] = S[r * 65 + i];
}

// ---------------------------------------------------------------------------------------------
// trsm_rows: X <- X * L11^-T, 64 rows per workgroup (one wavefront), lane r owns row r in registers.
//   x_k = b_k / L_kk ;  b_j -= x_k * L_jk  (j > k)      — right-looking, full ILP over j.
// L11 is read from LDS as wave-uniform broadcasts; 1 / L_kk comes from `invdiag` (written by potf2).
// Measured alternatives that were SLOWER on MI355X (profiles/r01_notes): L through the scalar cache
// (s_load_dwordx16 + SGPR operands: 55 us — 313 waves x 232 dependent loads of the same 16 KiB) and
// a transposed LDS image shared by two waves (39 us: the transposing store is a 64-way bank conflict).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void trsm_rows_kernel(T* __restrict__ X, int64_t ldx, const T* __restrict__ L11,
                                                       int64_t ldl, const T* __restrict__ invd_g, int64_t M,
                                                       const int* __restrict__ info) {
    if (info && *info != 0) return;
    __shared__ T SL[64 * 64];   // L11, row-major, read uniformly
    __shared__ T SX[64 * 65];   // X tile (transposition buffer)
    __shared__ T invd[64];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    for (int r = 0; r < 64; ++r) SL[r * 64 + t] = L11[(int64_t)r * ldl + t];
    for (int r = 0; r < 64; ++r) {
        int64_t gr = row0 + r;
        gr = gr < M ? gr : M - 1;
        SX[r * 65 + t] = X[gr * ldx + t];
    }
    invd[t] = invd_g[t];
    __syncthreads();
    T b[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) b[c] = SX[t * 65 + c];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const T xk = b[k] * invd[k];
        b[k] = xk;
#pragma unroll
        for (int j = k + 1; j < 64; ++j) b[j] -= xk * SL[j * 64 + k];
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) SX[t * 65 + c] = b[c];
    __syncthreads();
    for (int r = 0; r < 64; ++r) {
        const int64_t gr = row0 + r;
        if (gr < M) X[gr * ldx + t] = SX[r * 65 + t];
    }
}

// ---------------------------------------------------------------------------------------------
// bsolve_step: every workgroup redundantly back-substitutes the 64 x 64 diagonal block in its
// first wavefront (about 1 us, removes a launch from the critical path), workgroup 0 publishes
// alpha_b, then all workgroups apply  z[j] -= sum_i L[j0+i][j] * alpha_b[i]  to their 256 columns.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bsolve_step_kernel(const T* __restrict__ Arow, int64_t ld, int64_t j0,
                                                          T* __restrict__ z, T* __restrict__ alpha) {
    __shared__ T SL[64 * 65];
    __shared__ T sal[64];
    const int tid = threadIdx.x;
    const T* Lbb = Arow + j0;  // Arow = row j0 of the factor
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        SL[r * 65 + c] = Lbb[(int64_t)r * ld + c];
    }
    __syncthreads();
    if (tid < 64) {
        T zj = z[j0 + tid];
        const T invd = T(1) / SL[tid * 65 + tid];
        T aj = T(0);
        for (int i = 63; i >= 0; --i) {
            // alpha_i = z_i / L_ii  (z_i already holds the fully updated right-hand side)
            const T ai = __shfl(zj * invd, i, 64);
            if (tid == i) aj = ai;
            if (tid < i) zj -= SL[i * 65 + tid] * ai;  // row i of L, columns < i
        }
        sal[tid] = aj;
        if (blockIdx.x == 0) alpha[j0 + tid] = aj;
    }
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * 256 + tid;
    if (j < j0) {
        const T* col = Arow + j;
        T acc = T(0);
#pragma unroll 8
        for (int i = 0; i < 64; ++i) acc += col[(int64_t)i * ld] * sal[i];
        z[j] -= acc;
    }
}

// ---------------------------------------------------------------------------------------------
// deterministic single-workgroup reductions (fixed stride order + fixed tree)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

template <typename T>
__global__ __launch_bounds__(1024) void finalize_kernel(const T* __restrict__ A, int64_t ld, int64_t n,
                                                        const T* __restrict__ y, const T* __restrict__ alpha,
                                                        double* __restrict__ out) {
    __shared__ double sh[1024];
    double sl = 0.0, sd = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        sl += log((double)A[i * ld + i]);
        sd += (double)y[i] * (double)alpha[i];
    }
    const double logdiag = block_sum_1024(sl, sh);
    const double dot = block_sum_1024(sd, sh);
    if (threadIdx.x == 0) {
        const double logdet = 2.0 * logdiag;                    // PDMats.logdet = 2 sum log U_ii
        out[0] = -(dot + logdet + 1.8378770664093453 * (double)n) / 2.0;  // GPE.jl:210
        out[1] = logdet;
        out[2] = dot;
    }
}

// sum_i log A[i][col_off + i]  (a shard's share of logdet / 2)
template <typename T>
__global__ __launch_bounds__(1024) void logdiag_kernel(const T* __restrict__ A, int64_t ld, int64_t nrows, int64_t col_off,
                                                       double* __restrict__ out) {
    __shared__ double sh[1024];
    double sl = 0.0;
    for (int64_t i = threadIdx.x; i < nrows; i += 1024) sl += log((double)A[i * ld + col_off + i]);
    const double r = block_sum_1024(sl, sh);
    if (threadIdx.x == 0) out[0] = r;
}

template <typename T>
__global__ __launch_bounds__(256) void row_gemv_kernel(const T* __restrict__ R, int64_t ldr, int64_t n,
                                                       const T* __restrict__ alpha, const T* __restrict__ mean,
                                                       T* __restrict__ mu) {
    __shared__ double sh[256];
    const int64_t p = blockIdx.x;
    const T* row = R + p * ldr;
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) s += (double)row[j] * (double)alpha[j];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) mu[p] = (T)((double)mean[p] + sh[0]);  // GP.jl:26
}

template <typename T>
__global__ __launch_bounds__(256) void row_var_kernel(const T* __restrict__ R, int64_t ldr, int64_t n, double kdiag,
                                                      T* __restrict__ var) {
    __shared__ double sh[256];
    const int64_t p = blockIdx.x;
    const T* row = R + p * ldr;
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) {
        const double v = (double)row[j];
        s += v * v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double v = kdiag - sh[0];
        var[p] = (T)(v > 0.0 ? v : 0.0);  // GP.jl:75  max(diag(sig)[1], 0.0)
    }
}

}  // namespace

template <typename T>
void launch_potf2(gpmi_ctx* ctx, T* A, int64_t ld, T* invdiag, int* info, int64_t pivot_base) {
    ProfScope ps(ctx, GPMI_PROF_PANEL, 64.0 * 64.0 * 64.0 / 3.0);
    hipLaunchKernelGGL(potf2_kernel<T>, dim3(1), dim3(64), 0, ctx->stream, A, ld, invdiag, info, pivot_base);
}
template <typename T>
void launch_trsm_rows(gpmi_ctx* ctx, T* X, int64_t ldx, const T* L11, int64_t ldl, const T* invdiag, int64_t M,
                      const int* info) {
    if (M <= 0) return;
    ProfScope ps(ctx, GPMI_PROF_PANEL, (double)M * 64.0 * 64.0);
    hipLaunchKernelGGL(trsm_rows_kernel<T>, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, ctx->stream, X, ldx, L11, ldl,
                       invdiag, M, info);
}
template <typename T>
void launch_bsolve_step(gpmi_ctx* ctx, const T* Arow, int64_t ld, int64_t j0, T* z, T* alpha) {
    const unsigned blocks = (unsigned)(j0 > 0 ? (j0 + 255) / 256 : 1);
    hipLaunchKernelGGL(bsolve_step_kernel<T>, dim3(blocks), dim3(256), 0, ctx->stream, Arow, ld, j0, z, alpha);
}
template <typename T>
void launch_finalize(gpmi_ctx* ctx, const T* A, int64_t ld, int64_t n, const T* y, const T* alpha, double* out) {
    hipLaunchKernelGGL(finalize_kernel<T>, dim3(1), dim3(1024), 0, ctx->stream, A, ld, n, y, alpha, out);
}
template <typename T>
void launch_row_gemv(gpmi_ctx* ctx, const T* R, int64_t ldr, int64_t P, int64_t n, const T* alpha, const T* mean,
                     T* mu) {
    if (P <= 0) return;
    hipLaunchKernelGGL(row_gemv_kernel<T>, dim3((unsigned)P), dim3(256), 0, ctx->stream, R, ldr, n, alpha, mean, mu);
}
template <typename T>
void launch_row_var(gpmi_ctx* ctx, const T* R, int64_t ldr, int64_t P, int64_t n, double kdiag, T* var) {
    if (P <= 0) return;
    hipLaunchKernelGGL(row_var_kernel<T>, dim3((unsigned)P), dim3(256), 0, ctx->stream, R, ldr, n, kdiag, var);
}

template <typename T>
void launch_logdiag(gpmi_ctx* ctx, const T* A, int64_t ld, int64_t nrows, int64_t col_off, double* out) {
    hipLaunchKernelGGL(logdiag_kernel<T>, dim3(1), dim3(1024), 0, ctx->stream, A, ld, nrows, col_off, out);
}

#define INST(T)                                                                                                   \
    template void launch_potf2<T>(gpmi_ctx*, T*, int64_t, T*, int*, int64_t);                                     \
    template void launch_trsm_rows<T>(gpmi_ctx*, T*, int64_t, const T*, int64_t, const T*, int64_t, const int*);  \
    template void launch_bsolve_step<T>(gpmi_ctx*, const T*, int64_t, int64_t, T*, T*);                           \
    template void launch_finalize<T>(gpmi_ctx*, const T*, int64_t, int64_t, const T*, const T*, double*);         \
    template void launch_row_gemv<T>(gpmi_ctx*, const T*, int64_t, int64_t, int64_t, const T*, const T*, T*);     \
    template void launch_row_var<T>(gpmi_ctx*, const T*, int64_t, int64_t, int64_t, double, T*);                  \
    template void launch_logdiag<T>(gpmi_ctx*, const T*, int64_t, int64_t, int64_t, double*);
INST(double)
INST(float)

}  // namespace gpmi
