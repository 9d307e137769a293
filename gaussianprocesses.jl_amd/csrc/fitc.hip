// fitc.hip — FITC sparse approximation on the device (BASELINE.json configs[4]).
//
// Reference: update_cK!(::FullyIndepPDMat, …)  src/sparse/fully_indep_train_conditional.jl:134-156,
//            `\` :38-41, logdet :80, get_alpha_u :279-286, predictMVN :321-329 (= DTC, determ_train_conditional.jl:41-59,
//            on top of SoR, subsetofregressors.jl:303-321), update_mll! src/GPE.jl:202-212.
//
// With m inducing points and n observations, in the coordinates whitened by Kuu + 1e-10 I = Luu Luu':
//   W = Kfu Luu^-T  (n x m, one observation per row)   cov_kernel + whiten_rows (rows64 with refinement + update GEMMs)
//   Lambda_i = sigma^2 + k(x_i, x_i) - |W_i|^2         one pass over W                                   (:146-148)
//   U' = W' Lambda^-1/2  (m x n: the layout whose K dimension is contiguous)   tiled transpose + scale
//   B = I + U' U'' + 1e-10 (Luu' Luu)^-1               = Luu^-1 (SigmaQR + 1e-10 I) Luu^-T               (:150-153)
//        U' U'' is a split-K SYRK: K = n is cut into S chunks with separate partial outputs in ONE launch (the 528
//        lower tiles of a 4096^2 matrix do not fill 512 workgroup slots twice; 16 x 528 work items do); n m^2 flops
//   t = U' (Lambda^-1/2 r), carried as an extra row through chol(B) -> z, back-substitution -> c = B^-1 t
//   alpha = Lambda^-1 (r - W c)                        the Woodbury solve of :38-41
//   alpha_u = Luu^-T c                                 = SigmaQR^-1 Kuf Lambda^-1 r                       (:279-286)
//   mll = -(r'alpha + logdet B + sum log Lambda + n log 2 pi) / 2     (logdet SigmaQR - logdet Kuu = logdet B, :80)
// This is the reference's mathematics (both make_posdef! nuggets included) in the numerically stable order: SigmaQR
// itself has a condition number ~ n / (sigma^2 1e-10) and its small pivots are rounding noise in ANY fp64 Cholesky
// (LAPACK's included — tools/illcond_check.py, LABBOOK.md 3.6), while B is conditioned like 1 + n k / sigma^2.
// Kuu is regularised only by the nugget, so every product with a stored block inverse of Luu carries one step of
// iterative refinement (rows64, panel.hip).  Memory: two n x m matrices (W and U'), 32.8 GB each at n = 1e6, m = 4096.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "chol.h"
#include "common.h"

struct gpmi_fitc {
    gpmi_ctx* ctx = nullptr;
    int dtype = 64, d = 0;
    int64_t n = 0, npad = 0;  // observations; npad = n rounded up so that the split-K chunks are whole slabs
    int64_t m = 0, mpad = 0;  // inducing points; mpad = m rounded up to 64 (padding rows / cols: identity, zero)
    int64_t ldm = 0, ldn = 0; // leading dimensions of the m-wide and n-wide matrices: never a multiple of 4 KiB (a power-of-two
                              // row stride parks every row of a tile on the same HBM channels: 23 instead of 45 TFLOP/s)
    int nsplit = 1;           // K chunks of the SigmaQR product
    void *x = nullptr, *xu = nullptr;
    void *Auu = nullptr, *linv_uu = nullptr, *linv256_uu = nullptr, *invdiag_uu = nullptr;  // Kuu factor
    void *AS = nullptr, *linv_S = nullptr, *linv256_S = nullptr, *invdiag_S = nullptr;      // B = L_B L_B' (+ carried row)
    void *F = nullptr, *U = nullptr, *Cpart = nullptr, *G1 = nullptr, *G = nullptr;
    void *lam = nullptr, *rs = nullptr, *r = nullptr, *alpha = nullptr, *au = nullptr, *cvec = nullptr, *tmp = nullptr;
    double* part = nullptr;  // reduction partials
    // gradient scratch (gpmi_fitc_grad), allocated on first use: two n x m buffers, five m x m matrices, vectors, partials
    void *gA = nullptr, *gB = nullptr, *gM[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}, *gq = nullptr, *gv = nullptr, *gbt = nullptr;
    double* gpart = nullptr;
    int64_t gpart_cap = 0;
    void *rows = nullptr, *xp = nullptr, *small = nullptr;  // predict scratch
    int64_t rows_cap = 0, xp_cap = 0, small_cap = 0;
    bool fitted = false;
    double mll = 0.0;
};

namespace gpmi {
namespace {

constexpr int RED_BLOCKS = 1024;

// Lambda_i = noise + kdiag - sum_j W[i][j]^2 ;  rs_i = Lambda_i^-1/2.  A non-positive Lambda latches info = i + 1.
template <typename T>
__global__ __launch_bounds__(256) void fitc_lambda_kernel(const T* __restrict__ W, int64_t ldw, int64_t m, double noise,
                                                          double kdiag, T* __restrict__ lam, T* __restrict__ rs,
                                                          int* __restrict__ info) {
    __shared__ double sh[256];
    const int64_t i = blockIdx.x;
    const T* row = W + i * ldw;
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < m; j += 256) {
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
        const double l = noise + kdiag - sh[0];
        lam[i] = (T)l;
        rs[i] = (T)(1.0 / sqrt(l));
        if (!(l > 0.0)) atomicCAS(info, 0, (int)(i < 2147483000 ? i + 1 : 2147483000));
    }
}

// out[a][i] = in[i][a] * scale[i]  (scale == nullptr: plain transpose); in: rows x cols (ld_in), out: cols x ld_out.
// 32 x 32 tiles through LDS so that both sides are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void fitc_transpose_kernel(const T* __restrict__ in, int64_t ld_in, int64_t rows, int64_t cols,
                                                             const T* __restrict__ scale, T* __restrict__ out, int64_t ld_out) {
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t i0 = (int64_t)blockIdx.x * 32, a0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = i0 + ty + 8 * q, a = a0 + tx;
        T v = T(0);
        if (i < rows && a < cols) {
            v = in[i * ld_in + a];
            if (scale) v *= scale[i];
        }
        tile[ty + 8 * q][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t a = a0 + ty + 8 * q, i = i0 + tx;
        if (a < cols && i < rows) out[a * ld_out + i] = tile[tx][ty + 8 * q];
    }
}

// t[a] = sum_i U'[a][i] rs[i] r[i]   (one workgroup per inducing point; fixed summation tree)
template <typename T>
__global__ __launch_bounds__(256) void fitc_gemv_rows_kernel(const T* __restrict__ U, int64_t ldu, int64_t n,
                                                             const T* __restrict__ rs, const T* __restrict__ r, T* __restrict__ t) {
    __shared__ double sh[256];
    const T* row = U + (int64_t)blockIdx.x * ldu;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)row[i] * ((double)rs[i] * (double)r[i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) t[blockIdx.x] = (T)sh[0];
}

// B[i][j] = [i == j] + sum_s part[s][i][j] + eps * GG[i][j]   (lower triangle; fixed order over s)
template <typename T>
__global__ __launch_bounds__(256) void fitc_assemble_b_kernel(T* __restrict__ B, int64_t ld, const T* __restrict__ part, int nsplit,
                                                              int64_t stride, const T* __restrict__ GG, double eps) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > i) return;
    T acc = T(0);
    for (int s = 0; s < nsplit; ++s) acc += part[(int64_t)s * stride + i * ld + j];
    B[i * ld + j] = (i == j ? T(1) : T(0)) + acc + (T)eps * GG[i * ld + j];
}

// alpha[i] = (r[i] - (sum_a U'[a][i] c[a]) / rs[i]) / lam[i]     (W c through the scaled layout)
template <typename T>
__global__ __launch_bounds__(256) void fitc_alpha_kernel(const T* __restrict__ U, int64_t ldu, int64_t m, int64_t n,
                                                         const T* __restrict__ au, const T* __restrict__ rs,
                                                         const T* __restrict__ lam, const T* __restrict__ r, T* __restrict__ alpha) {
    __shared__ T sau[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
    for (int64_t a_base = 0; a_base < m; a_base += 256) {
        __syncthreads();
        sau[threadIdx.x] = (a_base + threadIdx.x < m) ? au[a_base + threadIdx.x] : T(0);
        __syncthreads();
        if (i < n) {
            const int64_t cnt = (m - a_base < 256) ? (m - a_base) : 256;
            const T* col = U + a_base * ldu + i;
            int64_t a = 0;
            for (; a + 3 < cnt; a += 4) {
                a0 += col[a * ldu] * sau[a];
                a1 += col[(a + 1) * ldu] * sau[a + 1];
                a2 += col[(a + 2) * ldu] * sau[a + 2];
                a3 += col[(a + 3) * ldu] * sau[a + 3];
            }
            for (; a < cnt; ++a) a0 += col[a * ldu] * sau[a];
        }
    }
    if (i < n) alpha[i] = (r[i] - ((a0 + a1) + (a2 + a3)) / rs[i]) / lam[i];
}

// partial sums of r'alpha and sum log Lambda (RED_BLOCKS workgroups, fixed trees), then one workgroup finishes
template <typename T>
__global__ __launch_bounds__(256) void fitc_sums_kernel(const T* __restrict__ r, const T* __restrict__ alpha,
                                                        const T* __restrict__ lam, int64_t n, double* __restrict__ part) {
    __shared__ double sh0[256], sh1[256];
    double d = 0.0, l = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        d += (double)r[i] * (double)alpha[i];
        l += log((double)lam[i]);
    }
    sh0[threadIdx.x] = d;
    sh1[threadIdx.x] = l;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            sh0[threadIdx.x] += sh0[threadIdx.x + k];
            sh1[threadIdx.x] += sh1[threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = sh0[0];
        part[2 * blockIdx.x + 1] = sh1[0];
    }
}
// out[0] = mll, out[1] = logdet, out[2] = r'alpha  from the partials and sum log diag(L_B) (logdiag[0])
__global__ __launch_bounds__(256) void fitc_finish_kernel(const double* __restrict__ part, int nblocks,
                                                          const double* __restrict__ logdiag, int64_t n, double* __restrict__ out) {
    __shared__ double sh0[256], sh1[256];
    double d = 0.0, l = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        d += part[2 * b];
        l += part[2 * b + 1];
    }
    sh0[threadIdx.x] = d;
    sh1[threadIdx.x] = l;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            sh0[threadIdx.x] += sh0[threadIdx.x + k];
            sh1[threadIdx.x] += sh1[threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double logdet = 2.0 * logdiag[0] + sh1[0];  // logdet SigmaQR - logdet Kuu + logdet Lambda, fully_indep…:80
        out[0] = -(sh0[0] + logdet + 1.8378770664093453 * (double)n) / 2.0;  // GPE.jl:210
        out[1] = logdet;
        out[2] = sh0[0];
    }
}

// var[p] = max(kdiag - |V1_p|^2 + |V2_p|^2, 0)   (DTC predictive variance, determ_train_conditional.jl:56; clamp GP.jl:75)
template <typename T>
__global__ __launch_bounds__(256) void fitc_var_kernel(const T* __restrict__ V1, const T* __restrict__ V2, int64_t ld, int64_t m,
                                                       double kdiag, T* __restrict__ var) {
    __shared__ double sh[256];
    const T* a = V1 + (int64_t)blockIdx.x * ld;
    const T* b = V2 + (int64_t)blockIdx.x * ld;
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < m; j += 256) {
        const double u = (double)a[j], v = (double)b[j];
        s += v * v - u * u;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double v = kdiag + sh[0];
        var[blockIdx.x] = (T)(v > 0.0 ? v : 0.0);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fitc_negate_kernel(T* __restrict__ A, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) A[i] = -A[i];
}

// ---- gradient (gpmi_fitc_grad) -------------------------------------------------------------------------------------
// out[i][a] = in[i][a] * s[i]
template <typename T>
__global__ __launch_bounds__(256) void fitc_scale_rows_kernel(const T* __restrict__ in, int64_t ld, int64_t n, int64_t m,
                                                              const T* __restrict__ s, T* __restrict__ out) {
    const int64_t i = blockIdx.y;
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a < m) out[i * ld + a] = in[i * ld + a] * s[i];
}
// q[i] = alpha_i^2 - (Sigma^-1)_ii,  (Sigma^-1)_ii = (1 - |g_i|^2) / Lambda_i,  g_i = L_B^-1 u'_i  (row i of G);  v[i] = Lambda_i alpha_i
template <typename T>
__global__ __launch_bounds__(256) void fitc_q_kernel(const T* __restrict__ G, int64_t ld, int64_t m, const T* __restrict__ alpha,
                                                     const T* __restrict__ lam, T* __restrict__ q, T* __restrict__ v) {
    __shared__ double sh[256];
    const int64_t i = blockIdx.x;
    const T* row = G + i * ld;
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < m; j += 256) {
        const double g = (double)row[j];
        s += g * g;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double a = (double)alpha[i], l = (double)lam[i];
        q[i] = (T)(a * a - (1.0 - sh[0]) / l);
        v[i] = (T)(l * a);
    }
}
// Ft[i][a] = alpha_i bt[a] - R2[i][a] rs_i - q_i W[i][a]      (in place on R2)
template <typename T>
__global__ __launch_bounds__(256) void fitc_ftilde_kernel(T* __restrict__ R2, const T* __restrict__ W, int64_t ld, int64_t m,
                                                          const T* __restrict__ alpha, const T* __restrict__ bt,
                                                          const T* __restrict__ rs, const T* __restrict__ q) {
    const int64_t i = blockIdx.y;
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a < m) R2[i * ld + a] = alpha[i] * bt[a] - R2[i * ld + a] * rs[i] - q[i] * W[i * ld + a];
}
// out[a][i] = U'[a][i] q_i Lambda_i   (so that out U'' = W' diag(q) W)
template <typename T>
__global__ __launch_bounds__(256) void fitc_scale_cols_kernel(const T* __restrict__ U, int64_t ld, int64_t n, int64_t npad,
                                                              const T* __restrict__ q, const T* __restrict__ lam, T* __restrict__ out) {
    const int64_t a = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < npad) out[a * ld + i] = i < n ? U[a * ld + i] * q[i] * lam[i] : T(0);
}
// Ht = -bt bt' + I - Binv + sum_s part[s]   (full symmetric m x m from the lower triangles)
template <typename T>
__global__ __launch_bounds__(256) void fitc_htilde_kernel(T* __restrict__ H, int64_t ld, const T* __restrict__ bt,
                                                          const T* __restrict__ Binv, const T* __restrict__ part, int nsplit,
                                                          int64_t stride) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > i) return;
    T acc = T(0);
    for (int s = 0; s < nsplit; ++s) acc += part[(int64_t)s * stride + i * ld + j];
    const T h = -bt[i] * bt[j] + (i == j ? T(1) : T(0)) - Binv[i * ld + j] + acc;
    H[i * ld + j] = h;
    H[j * ld + i] = h;
}
// out[0] = sum_i q_i  (fixed tree)
template <typename T>
__global__ __launch_bounds__(1024) void fitc_sumq_kernel(const T* __restrict__ q, int64_t n, double* __restrict__ out) {
    __shared__ double sh[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)q[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// factor an mpad x mpad covariance held in A (lower, identity padding) + its 256-block inverses
template <typename T>
void factor_dense(gpmi_ctx* c, T* A, int64_t ld, int64_t mpad, int64_t extra, T* linv, T* linv256, T* invdiag) {
    c->refine_solves = true;  // the only regularisation of Kuu / SigmaQR is make_posdef!'s 1e-10 nugget
    (void)cholesky_lower<T>(c, A, ld, linv, invdiag, mpad, extra, c->d_info);  // refinement on: no scratch, cannot fail
    c->refine_solves = c->refine_default;
    launch_linv256<T>(c, A, ld, linv, linv256, mpad, c->d_info);
}

template <typename T>
int fitc_fit_t(gpmi_fitc* f, const gpmi_kernel* k, double log_noise, const void* y_minus_mu, double* mll_out, void* alpha_out,
               int64_t* info_out) {
    gpmi_ctx* c = f->ctx;
    const int64_t n = f->n, npad = f->npad, m = f->m, mpad = f->mpad, ldm = f->ldm, ldn = f->ldn;
    f->fitted = false;
    la_reset(c);
    int rc = upload_program(c, k, f->d);
    if (rc != GPMI_OK) return rc;
    const double kdiag = c->h_prog->kdiag;
    const double noise = exp(2.0 * log_noise);
    T *Auu = (T*)f->Auu, *AS = (T*)f->AS, *F = (T*)f->F, *U = (T*)f->U;
    GPMI_HIP(c, hipMemsetAsync(c->d_info, 0, sizeof(int), c->stream));
    GPMI_HIP(c, hipMemcpyAsync(f->r, y_minus_mu, (size_t)n * sizeof(T), hipMemcpyHostToDevice, c->stream));

    // Kuu + 1e-10 I = Luu Luu'  (fully_indep…:139-141)
    launch_cov<T>(c, (const T*)f->xu, m, (const T*)f->xu, m, f->d, Auu, ldm, mpad, mpad, COV_LOWER | COV_NUGGET | COV_PAD_IDENTITY,
                  1e-10, nullptr);
    factor_dense<T>(c, Auu, ldm, mpad, 0, (T*)f->linv_uu, (T*)f->linv256_uu, (T*)f->invdiag_uu);
    // Kfu (:142), whitened in place: W = Kfu Luu^-T
    launch_cov<T>(c, (const T*)f->x, n, (const T*)f->xu, m, f->d, F, ldm, n, mpad, 0, 0.0, nullptr);
    {
        ProfScope ps(c, GPMI_PROF_SOLVE, (double)n * (double)mpad * (double)mpad);
        c->refine_solves = true;
        whiten_rows<T>(c, Auu, ldm, (const T*)f->linv_uu, mpad, F, ldm, n);
        // G1 = Luu^-T (the same whitening of an identity), for the SigmaQR nugget in whitened coordinates
        launch_set_identity<T>(c, (T*)f->G1, ldm, mpad);
        whiten_rows<T>(c, Auu, ldm, (const T*)f->linv_uu, mpad, (T*)f->G1, ldm, mpad);
        c->refine_solves = c->refine_default;
    }
    // Qdiag (:147), Lambda (:148)
    hipLaunchKernelGGL(fitc_lambda_kernel<T>, dim3((unsigned)n), dim3(256), 0, c->stream, (const T*)F, ldm, mpad, noise, kdiag,
                       (T*)f->lam, (T*)f->rs, c->d_info);
    // U' = W' Lambda^-1/2 (padding columns zero) and G = G1' = Luu^-1
    if (npad > n)
        GPMI_HIP(c, hipMemset2DAsync(U + n, (size_t)ldn * sizeof(T), 0, (size_t)(npad - n) * sizeof(T), (size_t)mpad, c->stream));
    hipLaunchKernelGGL(fitc_transpose_kernel<T>, dim3((unsigned)((n + 31) / 32), (unsigned)(mpad / 32)), dim3(256), 0, c->stream,
                       (const T*)F, ldm, n, mpad, (const T*)f->rs, U, ldn);
    hipLaunchKernelGGL(fitc_transpose_kernel<T>, dim3((unsigned)(mpad / 32), (unsigned)(mpad / 32)), dim3(256), 0, c->stream,
                       (const T*)f->G1, ldm, mpad, mpad, (const T*)nullptr, (T*)f->G, ldm);
    // t = W' Lambda^-1 r  ->  the carried row of B
    hipLaunchKernelGGL(fitc_gemv_rows_kernel<T>, dim3((unsigned)mpad), dim3(256), 0, c->stream, (const T*)U, ldn, n,
                       (const T*)f->rs, (const T*)f->r, AS + mpad * ldm);
    // B = I + U' U'' + 1e-10 G G'
    {
        const int64_t kc = npad / f->nsplit;
        GemmBatch gb{f->nsplit, kc, kc, mpad * ldm};
        launch_gemm_shape<T>(c, (T*)f->Cpart, ldm, U, ldn, U, ldn, mpad, mpad, kc, TileShape{0, 0, 1, 0, 1, 0}, c->d_info,
                             GEMM_OVERWRITE, &gb);
        launch_gemm_shape<T>(c, (T*)f->G1, ldm, (const T*)f->G, ldm, (const T*)f->G, ldm, mpad, mpad, mpad, TileShape{0, 0, 1, 0, 1, 0},
                             c->d_info, GEMM_OVERWRITE);  // G G' (lower) overwrites G1
        hipLaunchKernelGGL(fitc_assemble_b_kernel<T>, dim3((unsigned)((mpad + 255) / 256), (unsigned)mpad), dim3(256), 0, c->stream,
                           AS, ldm, (const T*)f->Cpart, f->nsplit, mpad * ldm, (const T*)f->G1, 1e-10);
    }
    // chol(B) carries t -> z = L_B^-1 t; back-substitution -> cvec = B^-1 t
    if (const int rc_chol = cholesky_lower<T>(c, AS, ldm, (T*)f->linv_S, (T*)f->invdiag_S, mpad, 1, c->d_info)) return rc_chol;
    launch_linv256<T>(c, AS, ldm, (const T*)f->linv_S, (T*)f->linv256_S, mpad, c->d_info);
    for (int64_t k0 = (mpad - 1) / NB * NB; k0 >= 0; k0 -= NB)
        launch_bsolve256<T>(c, AS + k0 * ldm, ldm, k0, (int)std::min<int64_t>(NB, mpad - k0),
                            (const T*)f->linv256_S + (k0 / NB) * NB * NB, AS + mpad * ldm, (T*)f->cvec);
    hipLaunchKernelGGL(fitc_alpha_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const T*)U, ldn, m, n,
                       (const T*)f->cvec, (const T*)f->rs, (const T*)f->lam, (const T*)f->r, (T*)f->alpha);
    // alpha_u = Luu^-T cvec  (get_alpha_u, :279-286): a back-substitution against Luu on a copy of cvec
    GPMI_HIP(c, hipMemcpyAsync(f->tmp, f->cvec, (size_t)mpad * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    for (int64_t k0 = (mpad - 1) / NB * NB; k0 >= 0; k0 -= NB)
        launch_bsolve256<T>(c, Auu + k0 * ldm, ldm, k0, (int)std::min<int64_t>(NB, mpad - k0),
                            (const T*)f->linv256_uu + (k0 / NB) * NB * NB, (T*)f->tmp, (T*)f->au);
    // mll (GPE.jl:210 with the logdet of :80)
    launch_logdiag<T>(c, AS, ldm, mpad, 0, c->d_scal + 4);
    hipLaunchKernelGGL(fitc_sums_kernel<T>, dim3(RED_BLOCKS), dim3(256), 0, c->stream, (const T*)f->r, (const T*)f->alpha,
                       (const T*)f->lam, n, f->part);
    hipLaunchKernelGGL(fitc_finish_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)f->part, RED_BLOCKS,
                       (const double*)(c->d_scal + 4), n, c->d_scal);
    int h_info = 0;
    GPMI_HIP(c, hipMemcpyAsync(c->h_scal, c->d_scal, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GPMI_HIP(c, hipMemcpyAsync(&h_info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    GPMI_HIP(c, hipStreamSynchronize(c->stream));
    GPMI_HIP(c, hipGetLastError());
    if (h_info < 0) {  // chain.hip: a dependency wait ran into its bound
        c->err = "chain kernel: a dependency wait timed out (GPMI_CHAIN=0 selects the multi-launch chain)";
        return GPMI_EDEVICE;
    }
    if (info_out) *info_out = h_info;
    if (h_info != 0) {
        c->err = "matrix is not positive definite; Cholesky factorization failed";
        return GPMI_ENOTPD;
    }
    f->mll = c->h_scal[0];
    f->fitted = true;
    if (mll_out) *mll_out = f->mll;
    if (alpha_out) GPMI_HIP(c, hipMemcpy(alpha_out, f->alpha, (size_t)n * sizeof(T), hipMemcpyDeviceToHost));
    return GPMI_OK;
}

template <typename T>
int fitc_predict_t(gpmi_fitc* f, const gpmi_kernel* k, int64_t P, const void* xpred, const void* mean_pred, int full_cov,
                   void* mu_out, void* var_out) {
    gpmi_ctx* c = f->ctx;
    const int64_t m = f->m, mpad = f->mpad, ldm = f->ldm;
    la_reset(c);
    int rc = upload_program(c, k, f->d);
    if (rc != GPMI_OK) return rc;
    const double kdiag = c->h_prog->kdiag;
    if ((rc = grow(c, &f->rows, &f->rows_cap, 2 * P * ldm * (int64_t)sizeof(T))) != GPMI_OK) return rc;
    if ((rc = grow(c, &f->xp, &f->xp_cap, P * f->d * (int64_t)sizeof(T))) != GPMI_OK) return rc;
    if ((rc = grow(c, &f->small, &f->small_cap, 3 * P * (int64_t)sizeof(T))) != GPMI_OK) return rc;
    T* V1 = (T*)f->rows;   // Kxu rows, whitened in place against Luu:  |V1_p|^2 = Qxx
    T* V2 = V1 + P * ldm;  // a copy of V1 whitened against L_B:        |V2_p|^2 = Kxu SigmaQR^-1 Kux
    T* xp = (T*)f->xp;
    T* d_mean = (T*)f->small;
    T* d_mu = d_mean + P;
    T* d_var = d_mu + P;
    GPMI_HIP(c, hipMemcpyAsync(xp, xpred, (size_t)(P * f->d) * sizeof(T), hipMemcpyHostToDevice, c->stream));
    GPMI_HIP(c, hipMemcpyAsync(d_mean, mean_pred, (size_t)P * sizeof(T), hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, GPMI_PROF_PREDICT, 2.0 * (double)mpad * (double)mpad * (double)P);
        launch_cov<T>(c, xp, P, (const T*)f->xu, m, f->d, V1, ldm, P, mpad, 0, 0.0, nullptr);
        launch_row_gemv<T>(c, V1, ldm, P, m, (const T*)f->au, d_mean, d_mu);  // mu = m(x*) + Kxu alpha_u, subsetofregressors.jl:314
        c->refine_solves = true;
        whiten_rows<T>(c, (const T*)f->Auu, ldm, (const T*)f->linv_uu, mpad, V1, ldm, P);
        c->refine_solves = c->refine_default;
        GPMI_HIP(c, hipMemcpyAsync(V2, V1, (size_t)(P * ldm) * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
        whiten_rows<T>(c, (const T*)f->AS, ldm, (const T*)f->linv_S, mpad, V2, ldm, P);
        if (!full_cov)
            hipLaunchKernelGGL(fitc_var_kernel<T>, dim3((unsigned)P), dim3(256), 0, c->stream, (const T*)V1, (const T*)V2, ldm, mpad,
                               kdiag, d_var);
    }
    GPMI_HIP(c, hipMemcpyAsync(mu_out, d_mu, (size_t)P * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    if (!full_cov) {
        GPMI_HIP(c, hipMemcpyAsync(var_out, d_var, (size_t)P * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        GPMI_HIP(c, hipStreamSynchronize(c->stream));
        GPMI_HIP(c, hipGetLastError());
        return GPMI_OK;
    }
    // Sigma = Kxx - Qxx + Kxu SigmaQR^-1 Kux  (determ_train_conditional.jl:56):  -(( -Kxx + V1 V1' ... )) via two C -= A B'
    const int64_t ldp = (P + 63) / 64 * 64;
    T* Kpp = nullptr;
    GPMI_HIP(c, hipMalloc(&Kpp, (size_t)(P * ldp) * sizeof(T)));
    launch_cov<T>(c, xp, P, xp, P, f->d, Kpp, ldp, P, ldp, 0, 0.0, nullptr);
    launch_gemm_nt<T>(c, Kpp, ldp, V1, ldm, V1, ldm, P, P, mpad, 0, nullptr);  // Kxx - Qxx
    hipLaunchKernelGGL(fitc_negate_kernel<T>, dim3((unsigned)((P * ldp + 255) / 256)), dim3(256), 0, c->stream, Kpp, P * ldp);
    launch_gemm_nt<T>(c, Kpp, ldp, V2, ldm, V2, ldm, P, P, mpad, 0, nullptr);  // -(Kxx - Qxx) - V2 V2'
    hipLaunchKernelGGL(fitc_negate_kernel<T>, dim3((unsigned)((P * ldp + 255) / 256)), dim3(256), 0, c->stream, Kpp, P * ldp);
    hipError_t e = hipMemcpy2DAsync(var_out, (size_t)P * sizeof(T), Kpp, (size_t)ldp * sizeof(T), (size_t)P * sizeof(T), (size_t)P,
                                    hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(Kpp);
    GPMI_HIP(c, e);
    GPMI_HIP(c, hipGetLastError());
    return GPMI_OK;
}

}  // namespace
// ---- update_dmll! on a FITC model: dmll_kern! (fully_indep_train_conditional.jl:200-234 over the SoR part,
// subsetofregressors.jl:219-256) and dmll_noise (:243-257), in the whitened coordinates of the fit.  With
//   b~ = W' alpha,  G = Lambda^-1/2 W L_B^-T,  (Sigma^-1)_ii = (1 - |g_i|^2) / Lambda_i,  q_i = alpha_i^2 - (Sigma^-1)_ii
//   F~ = alpha b~' - Lambda^-1/2 (G L_B^-1) - diag(q) W                (n x m),      F = F~ Luu^-1
//   H~ = -b~ b~' + (I - B^-1) + W' diag(q) W                            (m x m),      H = Luu^-T H~ Luu^-1
// the reference's V - T sums collapse to
//   dmll/dθ = <dKfu/dθ, F> + 1/2 <dKuu/dθ, H> + 1/2 (sum_i q_i) dk(x,x)/dθ ,      dmll/dlogNoise = sigma^2 sum_i q_i
// (checked against the literal restatement oracle.gp_oracle.fitc_update_dmll, itself checked by central differences).
// Five n m^2 products on gemm_nt_kernel — the m x m triangular inverses are formed explicitly (L_B^-1 by whitening an
// identity; Luu^-1 is the fit's G), so every step is a plain NT product — and three passes of the fused dK/dθ trace kernel
// (grad.hip, rectangular form): over (x, xu) with weights F, over (xu, xu) with weights H, and one pair at r = 0.
template <typename T>
int fitc_grad_t(gpmi_fitc* f, const gpmi_kernel* k, double log_noise, double* dkern_out, int32_t n_kern, double* dnoise_out) {
    gpmi_ctx* c = f->ctx;
    const int64_t n = f->n, npad = f->npad, m = f->m, mpad = f->mpad, ldm = f->ldm, ldn = f->ldn;
    la_reset(c);
    int rc = upload_program(c, k, f->d);
    if (rc != GPMI_OK) return rc;
    const int n_hyp = c->h_prog->n_hyp;
    if (n_kern != n_hyp) {
        c->err = "gpmi_fitc_grad: n_kern does not match the kernel's number of hyper-parameters";
        return GPMI_EARG;
    }
    if (n_hyp > GPMI_GRAD_MAX_PARAMS) {
        c->err = "gpmi_fitc_grad: more than GPMI_GRAD_MAX_PARAMS (5000) kernel hyper-parameters: the trace kernel's per-wave table does not fit the LDS";
        return GPMI_EARG;
    }
    // The per-point term sum_i q_i dk(x_i, x_i)/dtheta of fully_indep_train_conditional.jl:218 is evaluated as
    // (sum_i q_i) * dk/dtheta at r = 0: valid because every leaf of include/gpmi.h is STATIONARY (k(x, x) does not depend
    // on x).  A non-stationary leaf (Lin, Poly, ...) added to gpmi_op must be weighted per point here: refuse it until then.
    for (int o = 0; o < c->h_prog->n_ops; ++o) {
        const int op = c->h_prog->leaf[o].op;
        if (!((op >= GPMI_K_SE_ISO && op <= GPMI_K_CONST) || op == GPMI_K_SUM || op == GPMI_K_PROD)) {
            c->err = "gpmi_fitc_grad: the diagonal term assumes stationary leaves; this kernel has a leaf outside that set";
            return GPMI_EARG;
        }
    }
    const size_t es = sizeof(T);
    const int64_t big = std::max<int64_t>(n * ldm, mpad * ldn);
    if (!f->gA) GPMI_HIP(c, hipMalloc(&f->gA, (size_t)big * es));
    if (!f->gB) GPMI_HIP(c, hipMalloc(&f->gB, (size_t)big * es));
    for (auto& p : f->gM)
        if (!p) GPMI_HIP(c, hipMalloc(&p, (size_t)(mpad * ldm) * es));
    if (!f->gq) GPMI_HIP(c, hipMalloc(&f->gq, (size_t)n * es));
    if (!f->gv) GPMI_HIP(c, hipMalloc(&f->gv, (size_t)n * es));
    if (!f->gbt) GPMI_HIP(c, hipMalloc(&f->gbt, (size_t)mpad * es));
    const int64_t nb1 = ((n + 63) / 64) * ((m + 63) / 64), nb2 = ((m + 63) / 64) * ((m + 63) / 64);
    const int64_t need = (std::max(nb1, nb2) + 1) * (int64_t)(n_hyp + 1) * (int64_t)sizeof(double);
    if (f->gpart_cap < need) {
        if (f->gpart) hipFree(f->gpart);
        f->gpart = nullptr;
        f->gpart_cap = 0;
        GPMI_HIP(c, hipMalloc(&f->gpart, (size_t)need));
        f->gpart_cap = need;
    }
    T *bufA = (T*)f->gA, *bufB = (T*)f->gB, *W = (T*)f->F, *U = (T*)f->U;
    T *GBT = (T*)f->gM[0], *GB = (T*)f->gM[1], *GTuu = (T*)f->gM[2], *M3 = (T*)f->gM[3], *M4 = (T*)f->gM[4];
    T *q = (T*)f->gq, *v = (T*)f->gv, *bt = (T*)f->gbt;
    const TileShape rect{0, 0, 0, 0, 1, 0}, lower{0, 0, 1, 0, 1, 0};
    const dim3 gm((unsigned)((mpad + 255) / 256), (unsigned)n), gmm((unsigned)(mpad / 32), (unsigned)(mpad / 32));
    std::vector<double> t1((size_t)n_hyp + 1), t2((size_t)n_hyp + 1), t3((size_t)n_hyp + 1);
    double sumq = 0.0;
    {
        ProfScope ps(c, GPMI_PROF_SOLVE, 8.0 * (double)n * (double)mpad * (double)mpad);
        // L_B^-T (rows of the whitened identity), L_B^-1, Luu^-T
        launch_set_identity<T>(c, GBT, ldm, mpad);
        whiten_rows<T>(c, (const T*)f->AS, ldm, (const T*)f->linv_S, mpad, GBT, ldm, mpad);
        hipLaunchKernelGGL(fitc_transpose_kernel<T>, gmm, dim3(256), 0, c->stream, (const T*)GBT, ldm, mpad, mpad, (const T*)nullptr, GB, ldm);
        hipLaunchKernelGGL(fitc_transpose_kernel<T>, gmm, dim3(256), 0, c->stream, (const T*)f->G, ldm, mpad, mpad, (const T*)nullptr, GTuu, ldm);
        // G = (Lambda^-1/2 W) L_B^-T ;  q, v = Lambda alpha ;  b~ = W' alpha = U' (rs . v)
        hipLaunchKernelGGL(fitc_scale_rows_kernel<T>, gm, dim3(256), 0, c->stream, (const T*)W, ldm, n, mpad, (const T*)f->rs, bufA);
        launch_gemm_shape<T>(c, bufB, ldm, bufA, ldm, GB, ldm, n, mpad, mpad, rect, nullptr, GEMM_OVERWRITE);
        hipLaunchKernelGGL(fitc_q_kernel<T>, dim3((unsigned)n), dim3(256), 0, c->stream, (const T*)bufB, ldm, mpad, (const T*)f->alpha,
                           (const T*)f->lam, q, v);
        hipLaunchKernelGGL(fitc_gemv_rows_kernel<T>, dim3((unsigned)mpad), dim3(256), 0, c->stream, (const T*)U, ldn, n, (const T*)f->rs,
                           (const T*)v, bt);
        hipLaunchKernelGGL(fitc_sumq_kernel<T>, dim3(1), dim3(1024), 0, c->stream, (const T*)q, n, c->d_scal + 6);
        // R2 = G L_B^-1 ;  F~ (in place) ;  F = F~ Luu^-1
        launch_gemm_shape<T>(c, bufA, ldm, bufB, ldm, GBT, ldm, n, mpad, mpad, rect, nullptr, GEMM_OVERWRITE);
        hipLaunchKernelGGL(fitc_ftilde_kernel<T>, gm, dim3(256), 0, c->stream, bufA, (const T*)W, ldm, mpad, (const T*)f->alpha, (const T*)bt,
                           (const T*)f->rs, (const T*)q);
        launch_gemm_shape<T>(c, bufB, ldm, bufA, ldm, GTuu, ldm, n, mpad, mpad, rect, nullptr, GEMM_OVERWRITE);
        // <dKfu, F>
        const int64_t b1 = launch_dmll_rect<T>(c, (const T*)f->x, n, (const T*)f->xu, m, f->d, (const T*)bufB, ldm, f->gpart, n_hyp);
        launch_reduce_partials(c, f->gpart, b1, n_hyp + 1, (double*)M3);
        GPMI_HIP(c, hipMemcpyAsync(t1.data(), M3, t1.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        // W' diag(q) W = (U' diag(q Lambda)) U''  (split-K, lower tiles), B^-1 = L_B^-T L_B^-1 (lower), H~, H
        hipLaunchKernelGGL(fitc_scale_cols_kernel<T>, dim3((unsigned)((npad + 255) / 256), (unsigned)mpad), dim3(256), 0, c->stream,
                           (const T*)U, ldn, n, npad, (const T*)q, (const T*)f->lam, bufA);
        {
            const int64_t kc = npad / f->nsplit;
            GemmBatch gb{f->nsplit, kc, kc, mpad * ldm};
            launch_gemm_shape<T>(c, (T*)f->Cpart, ldm, bufA, ldn, U, ldn, mpad, mpad, kc, lower, nullptr, GEMM_OVERWRITE, &gb);
        }
        launch_gemm_shape<T>(c, M3, ldm, GBT, ldm, GBT, ldm, mpad, mpad, mpad, lower, nullptr, GEMM_OVERWRITE);
        hipLaunchKernelGGL(fitc_htilde_kernel<T>, dim3((unsigned)((mpad + 255) / 256), (unsigned)mpad), dim3(256), 0, c->stream, M4, ldm,
                           (const T*)bt, (const T*)M3, (const T*)f->Cpart, f->nsplit, mpad * ldm);
        launch_gemm_shape<T>(c, M3, ldm, M4, ldm, GTuu, ldm, mpad, mpad, mpad, rect, nullptr, GEMM_OVERWRITE);      // Y = H~ Luu^-1
        hipLaunchKernelGGL(fitc_transpose_kernel<T>, gmm, dim3(256), 0, c->stream, (const T*)M3, ldm, mpad, mpad, (const T*)nullptr, M4, ldm);
        launch_gemm_shape<T>(c, M3, ldm, M4, ldm, GTuu, ldm, mpad, mpad, mpad, rect, nullptr, GEMM_OVERWRITE);      // H = Y' Luu^-1
        // <dKuu, H>  and  dk(x, x)/dθ (one pair at distance 0, weight 1)
        const int64_t b2 = launch_dmll_rect<T>(c, (const T*)f->xu, m, (const T*)f->xu, m, f->d, (const T*)M3, ldm, f->gpart, n_hyp);
        launch_reduce_partials(c, f->gpart, b2, n_hyp + 1, (double*)M4);
        GPMI_HIP(c, hipMemcpyAsync(t2.data(), M4, t2.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        launch_set_identity<T>(c, GB, ldm, 1);  // GB is free again: a 1 x 1 weight matrix [1]
        const int64_t b3 = launch_dmll_rect<T>(c, (const T*)f->xu, 1, (const T*)f->xu, 1, f->d, (const T*)GB, ldm, f->gpart, n_hyp);
        launch_reduce_partials(c, f->gpart, b3, n_hyp + 1, (double*)GBT);
        GPMI_HIP(c, hipMemcpyAsync(t3.data(), GBT, t3.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        GPMI_HIP(c, hipMemcpyAsync(&sumq, c->d_scal + 6, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    GPMI_HIP(c, hipStreamSynchronize(c->stream));
    GPMI_HIP(c, hipGetLastError());
    for (int p = 0; p < n_hyp; ++p) dkern_out[p] = t1[(size_t)p] + 0.5 * t2[(size_t)p] + 0.5 * sumq * t3[(size_t)p];
    if (dnoise_out) *dnoise_out = exp(2.0 * log_noise) * sumq;
    return GPMI_OK;
}

}  // namespace gpmi

extern "C" {

void gpmi_fitc_destroy(gpmi_fitc* f) {
    if (!f) return;
    if (f->ctx) {
        hipSetDevice(f->ctx->device);
        hipStreamSynchronize(f->ctx->stream);
    }
    void* ptrs[] = {f->x, f->xu, f->Auu, f->linv_uu, f->linv256_uu, f->invdiag_uu, f->AS, f->linv_S, f->linv256_S, f->invdiag_S,
                    f->F, f->U, f->Cpart, f->G1, f->G, f->lam, f->rs, f->r, f->alpha, f->au, f->cvec, f->tmp, f->part, f->rows, f->xp,
                    f->small, f->gA, f->gB, f->gM[0], f->gM[1], f->gM[2], f->gM[3], f->gM[4], f->gq, f->gv, f->gbt, f->gpart};
    for (void* p : ptrs)
        if (p) hipFree(p);
    delete f;
}

int gpmi_fitc_create(gpmi_ctx* c, int dtype, int d, int64_t n, const void* x, int64_t m, const void* xu, gpmi_fitc** out) {
    using namespace gpmi;
    if (!c) return GPMI_EARG;
    if (!out || !x || !xu || (dtype != 64 && dtype != 32) || d <= 0 || n <= 0 || m <= 0) {
        c->err = "gpmi_fitc_create: bad argument (dtype must be 64|32, 1 <= d <= 2^20, n >= 1, m >= 1)";
        return GPMI_EARG;
    }
    *out = nullptr;
    GPMI_HIP(c, hipSetDevice(c->device));
    gpmi_fitc* f = new gpmi_fitc();
    f->ctx = c;
    f->dtype = dtype;
    f->d = d;
    f->n = n;
    f->m = m;
    f->mpad = (m + IB - 1) / IB * IB;
    // split K = n into chunks of whole 64-element slabs, enough of them that the lower tiles of the m x m product fill
    // the chip several times over
    int ns = 1;
    while (ns < 16 && n / (2 * ns) >= 4096) ns *= 2;
    f->nsplit = ns;
    f->npad = (n + 64 * ns - 1) / (64 * ns) * (64 * ns);
    const size_t es = dtype == 64 ? 8 : 4;
    const int64_t mpad = f->mpad, npad = f->npad;
    auto odd_ld = [es](int64_t w) { return (w * (int64_t)es) % 4096 == 0 ? w + 64 : w; };
    f->ldm = odd_ld(mpad);
    f->ldn = odd_ld(npad);
    const int64_t ldm = f->ldm, ldn = f->ldn;
    const size_t mm = (size_t)((mpad + 8) * ldm) * es;
    hipError_t e = hipMalloc(&f->x, (size_t)(n * d) * es);
    auto alloc = [&](void** p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
    };
    alloc(&f->xu, (size_t)(m * d) * es);
    alloc(&f->Auu, mm);
    alloc(&f->AS, mm);
    alloc(&f->linv_uu, (size_t)(mpad * IB) * es);
    alloc(&f->linv_S, (size_t)(mpad * IB) * es);
    alloc(&f->linv256_uu, (size_t)((mpad + NB - 1) / NB * NB * NB) * es);
    alloc(&f->linv256_S, (size_t)((mpad + NB - 1) / NB * NB * NB) * es);
    alloc(&f->invdiag_uu, (size_t)mpad * es);
    alloc(&f->invdiag_S, (size_t)mpad * es);
    alloc(&f->F, (size_t)(n * ldm) * es);
    alloc(&f->U, (size_t)(mpad * ldn) * es);
    alloc(&f->Cpart, (size_t)((int64_t)ns * mpad * ldm) * es);
    alloc(&f->lam, (size_t)n * es);
    alloc(&f->rs, (size_t)n * es);
    alloc(&f->r, (size_t)n * es);
    alloc(&f->alpha, (size_t)n * es);
    alloc(&f->au, (size_t)mpad * es);
    alloc(&f->cvec, (size_t)mpad * es);
    alloc(&f->tmp, (size_t)mpad * es);
    alloc(&f->G1, (size_t)(mpad * ldm) * es);
    alloc(&f->G, (size_t)(mpad * ldm) * es);
    alloc((void**)&f->part, (size_t)(2 * RED_BLOCKS) * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(f->x, x, (size_t)(n * d) * es, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(f->xu, xu, (size_t)(m * d) * es, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset((char*)f->Auu + (size_t)(mpad * ldm) * es, 0, (size_t)(8 * ldm) * es);
    if (e == hipSuccess) e = hipMemset((char*)f->AS + (size_t)(mpad * ldm) * es, 0, (size_t)(8 * ldm) * es);
    if (e != hipSuccess) {
        c->err = std::string("gpmi_fitc_create: ") + hipGetErrorString(e);
        gpmi_fitc_destroy(f);
        return GPMI_EDEVICE;
    }
    *out = f;
    return GPMI_OK;
}

int gpmi_fitc_fit(gpmi_fitc* f, const gpmi_kernel* k, double log_noise, const void* y_minus_mu, double* mll_out, void* alpha_out,
                  int64_t* info_out) {
    if (!f) return GPMI_EARG;
    if (!k || !y_minus_mu) {
        f->ctx->err = "gpmi_fitc_fit: null argument";
        return GPMI_EARG;
    }
    hipSetDevice(f->ctx->device);
    return f->dtype == 64 ? gpmi::fitc_fit_t<double>(f, k, log_noise, y_minus_mu, mll_out, alpha_out, info_out)
                          : gpmi::fitc_fit_t<float>(f, k, log_noise, y_minus_mu, mll_out, alpha_out, info_out);
}

int gpmi_fitc_predict(gpmi_fitc* f, const gpmi_kernel* k, int64_t p, const void* xpred, const void* mean_pred, int full_cov,
                      void* mu_out, void* var_out) {
    if (!f) return GPMI_EARG;
    if (!k || !xpred || !mean_pred || !mu_out || !var_out || p <= 0) {
        f->ctx->err = "gpmi_fitc_predict: bad argument";
        return GPMI_EARG;
    }
    if (!f->fitted) {
        f->ctx->err = "gpmi_fitc_predict: no factorisation (call gpmi_fitc_fit first)";
        return GPMI_EARG;
    }
    hipSetDevice(f->ctx->device);
    return f->dtype == 64 ? gpmi::fitc_predict_t<double>(f, k, p, xpred, mean_pred, full_cov, mu_out, var_out)
                          : gpmi::fitc_predict_t<float>(f, k, p, xpred, mean_pred, full_cov, mu_out, var_out);
}

int gpmi_fitc_grad(gpmi_fitc* f, const gpmi_kernel* k, double log_noise, double* dkern_out, int32_t n_kern, double* dnoise_out) {
    if (!f) return GPMI_EARG;
    if (!k || !dkern_out) {
        f->ctx->err = "gpmi_fitc_grad: null argument";
        return GPMI_EARG;
    }
    if (!f->fitted) {
        f->ctx->err = "gpmi_fitc_grad: no factorisation (call gpmi_fitc_fit first)";
        return GPMI_EARG;
    }
    hipSetDevice(f->ctx->device);
    return f->dtype == 64 ? gpmi::fitc_grad_t<double>(f, k, log_noise, dkern_out, n_kern, dnoise_out)
                          : gpmi::fitc_grad_t<float>(f, k, log_noise, dkern_out, n_kern, dnoise_out);
}

int gpmi_fitc_alpha_u(gpmi_fitc* f, void* out) {
    if (!f || !out) return GPMI_EARG;
    if (!f->fitted) {
        f->ctx->err = "gpmi_fitc_alpha_u: no factorisation";
        return GPMI_EARG;
    }
    const size_t es = f->dtype == 64 ? 8 : 4;
    gpmi_ctx* c = f->ctx;
    GPMI_HIP(c, hipMemcpy(out, f->au, (size_t)f->m * es, hipMemcpyDeviceToHost));
    return GPMI_OK;
}

}  // extern "C"
