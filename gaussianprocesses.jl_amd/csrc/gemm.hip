// gemm.hip — C[M x N] -= A[M x K] * B[N x K]'  on the gfx950 matrix cores.
//
// This is the Cholesky trailing update (the dpotrf/dsyrk work behind make_posdef!,
// src/GP.jl:110) and the whiten! update of predict (src/GP.jl:27): in the row-major
// lower factor both operands have K contiguous ("NT" form), which is exactly the
// v_mfma_{f64,f32}_16x16x4 operand layout: lane l supplies A[row = l & 15][k = l >> 4].
//
// Structure per 256-thread workgroup (4 wavefronts, 2 x 2):
//   128 x 128 output tile, each wavefront a 64 x 64 sub-tile = 4 x 4 MFMA tiles whose
//   accumulators (128 VGPR for fp64) live in registers for the whole K loop;
//   K is streamed in 128-byte slabs (16 doubles / 32 floats) through a double-buffered,
//   16-byte-padded LDS image filled with coalesced 16-B loads (one cache line per row);
//   fragments are read with ds_read_b128 (E consecutive k per lane — the k permutation is
//   identical for A and B, so the dot products are unchanged);
//   2 workgroups per CU (launch_bounds(256, 2)): one group's epilogue / global loads hide
//   under the other's MFMA stream, which is the only busy pipe (64 MFMAs per slab per wave).
// Tile order is XCD-aware: block b runs on XCD b % 8, so each XCD gets a contiguous run of
// 8 x 8 super-tiles (64 concurrent tiles share 16 operand panels in that XCD's 4 MiB L2).
#include "common.h"

namespace gpmi {

namespace {

template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
    using Acc = double __attribute__((ext_vector_type(4)));
    using Vec = double __attribute__((ext_vector_type(2)));
    static constexpr int E = 2;
    static constexpr int BK = 16;
    static __device__ __forceinline__ Acc mma(double a, double b, Acc c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // v_mfma_f64_16x16x4_f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int row_of(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <>
struct Mfma<float> {
    using Acc = float __attribute__((ext_vector_type(4)));
    using Vec = float __attribute__((ext_vector_type(4)));
    static constexpr int E = 4;
    static constexpr int BK = 32;
    static __device__ __forceinline__ Acc mma(float a, float b, Acc c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // v_mfma_f32_16x16x4_f32 C/D map: col = lane & 15, row = 4 * (lane >> 4) + reg
    static __device__ __forceinline__ int row_of(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(T* __restrict__ C, int64_t ldc, const T* __restrict__ A,
                                                         int64_t lda, const T* __restrict__ B, int64_t ldb, int64_t M,
                                                         int64_t N, int64_t K, int lower, int ntm, int ntn, int nsj,
                                                         int64_t total_lin, const int* __restrict__ info) {
    using MF = Mfma<T>;
    using Vec = typename MF::Vec;
    using Acc = typename MF::Acc;
    constexpr int E = MF::E;
    constexpr int BK = MF::BK;
    constexpr int LDS_LD = BK + E;  // +16 B pad per row
    constexpr int BM = GEMM_BM, BN = GEMM_BN;

    if (info && *info != 0) return;  // an earlier pivot failed: the factorisation is abandoned

    // ---- XCD-aware tile decode -------------------------------------------------------------
    const int64_t b = blockIdx.x;
    const int64_t chunk = total_lin >> 3;
    const int64_t lin = (b & 7) * chunk + (b >> 3);
    const int64_t u = lin >> 6;
    const int w = (int)(lin & 63);
    int64_t SI, SJ;
    if (lower) {
        SI = (int64_t)((sqrt(8.0 * (double)u + 1.0) - 1.0) * 0.5);
        while ((SI + 1) * (SI + 2) / 2 <= u) ++SI;
        while (SI * (SI + 1) / 2 > u) --SI;
        SJ = u - SI * (SI + 1) / 2;
    } else {
        SI = u / nsj;
        SJ = u - SI * nsj;
    }
    const int64_t ti = SI * SUPER + (w >> 3);
    const int64_t tj = SJ * SUPER + (w & 7);
    if (ti >= ntm || tj >= ntn) return;
    if (lower && tj * BN > ti * BM + BM - 1) return;
    const int64_t m0 = ti * BM, n0 = tj * BN;

    __shared__ __attribute__((aligned(16))) T As[2][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) T Bs[2][BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int r16 = lane & 15, g = lane >> 4;

    // staging map: 4 x 16-B chunks of A and of B per thread per slab
    const T* ga[4];
    const T* gb[4];
    int so[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        const int row = c >> 3, kc = c & 7;
        int64_t ra = m0 + row;
        ra = ra < M ? ra : M - 1;
        int64_t rb = n0 + row;
        rb = rb < N ? rb : N - 1;
        ga[i] = A + ra * lda + kc * E;
        gb[i] = B + rb * ldb + kc * E;
        so[i] = row * LDS_LD + kc * E;
    }

    Acc acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = T(0);

    Vec ra_[4], rb_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra_[i] = *reinterpret_cast<const Vec*>(ga[i]);
        rb_[i] = *reinterpret_cast<const Vec*>(gb[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<Vec*>(&As[0][so[i]]) = ra_[i];
        *reinterpret_cast<Vec*>(&Bs[0][so[i]]) = rb_[i];
    }
    __syncthreads();

    const int nk = (int)(K / BK);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int64_t ko = (int64_t)(kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra_[i] = *reinterpret_cast<const Vec*>(ga[i] + ko);
                rb_[i] = *reinterpret_cast<const Vec*>(gb[i] + ko);
            }
        }
        const T* as = &As[cur][(wm * 64 + r16) * LDS_LD + g * E];
        const T* bs = &Bs[cur][(wn * 64 + r16) * LDS_LD + g * E];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            Vec af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const Vec*>(as + mi * 16 * LDS_LD + h * (BK / 2));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni] = *reinterpret_cast<const Vec*>(bs + ni * 16 * LDS_LD + h * (BK / 2));
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = MF::mma(af[mi][e], bf[ni][e], acc[mi][ni]);
        }
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<Vec*>(&As[cur ^ 1][so[i]]) = ra_[i];
                *reinterpret_cast<Vec*>(&Bs[cur ^ 1][so[i]]) = rb_[i];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: C -= acc ------------------------------------------------------------------
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = m0 + wm * 64 + mi * 16 + MF::row_of(lane, r);
            if (row < M) {
                T* crow = C + row * ldc + n0 + wn * 64 + r16;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int64_t col = n0 + wn * 64 + ni * 16 + r16;
                    if (col < N) crow[ni * 16] -= acc[mi][ni][r];
                }
            }
        }
    }
}

// ---- peak-rate micro-benchmark: back-to-back MFMAs on 4 independent accumulators ------------
template <typename T>
__global__ __launch_bounds__(256) void mfma_peak_kernel(T* out, int iters) {
    using MF = Mfma<T>;
    typename MF::Acc a0, a1, a2, a3;
    for (int r = 0; r < 4; ++r) a0[r] = a1[r] = a2[r] = a3[r] = T(0);
    T x = T(threadIdx.x & 7) * T(0.125), y = T(1.0) + T(threadIdx.x & 3) * T(1e-3);
    for (int i = 0; i < iters; ++i) {
        a0 = MF::mma(x, y, a0);
        a1 = MF::mma(y, x, a1);
        a2 = MF::mma(x, x, a2);
        a3 = MF::mma(y, y, a3);
    }
    T s = T(0);
    for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == T(-1.2345)) out[0] = s;  // keep the chain live
}

}  // namespace

template <typename T>
void launch_gemm_nt(gpmi_ctx* ctx, T* C, int64_t ldc, const T* A, int64_t lda, const T* B, int64_t ldb, int64_t M,
                    int64_t N, int64_t K, int lower, const int* info) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    const int ntm = (int)((M + GEMM_BM - 1) / GEMM_BM);
    const int ntn = (int)((N + GEMM_BN - 1) / GEMM_BN);
    const int nsi = (ntm + SUPER - 1) / SUPER;
    const int nsj = (ntn + SUPER - 1) / SUPER;
    int64_t nsuper = lower ? (int64_t)nsi * (nsi + 1) / 2 : (int64_t)nsi * nsj;
    const int64_t total_lin = nsuper * SUPER * SUPER;
    double entries = lower ? (0.5 * (double)N * ((double)N + 1.0) + (double)(M - N) * (double)N) : (double)M * (double)N;
    ProfScope ps(ctx, lower ? GPMI_PROF_SYRK : GPMI_PROF_PANEL, 2.0 * entries * (double)K);
    hipLaunchKernelGGL(gemm_nt_kernel<T>, dim3((unsigned)total_lin), dim3(256), 0, ctx->stream, C, ldc, A, lda, B, ldb,
                       M, N, K, lower, ntm, ntn, nsj, total_lin, info);
}

template void launch_gemm_nt<double>(gpmi_ctx*, double*, int64_t, const double*, int64_t, const double*, int64_t,
                                     int64_t, int64_t, int64_t, int, const int*);
template void launch_gemm_nt<float>(gpmi_ctx*, float*, int64_t, const float*, int64_t, const float*, int64_t, int64_t,
                                    int64_t, int64_t, int, const int*);

template <typename T>
int mfma_peak(gpmi_ctx* ctx, double* tflops) {
    T* d_out = nullptr;
    GPMI_HIP(ctx, hipMalloc(&d_out, 64));
    const int iters = 20000;
    const int blocks = 256 * 4;  // 4 workgroups (16 waves) per CU
    hipEvent_t e0, e1;
    GPMI_HIP(ctx, hipEventCreate(&e0));
    GPMI_HIP(ctx, hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_peak_kernel<T>, dim3(blocks), dim3(256), 0, ctx->stream, d_out, 100);  // warm-up
    GPMI_HIP(ctx, hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(mfma_peak_kernel<T>, dim3(blocks), dim3(256), 0, ctx->stream, d_out, iters);
    GPMI_HIP(ctx, hipEventRecord(e1, ctx->stream));
    GPMI_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    GPMI_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * 4.0 /*waves*/ * (double)iters * 4.0 /*mfma per iter*/ * 2.0 * 16 * 16 * 4;
    *tflops = flops / ((double)ms * 1e-3) / 1e12;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d_out);
    return GPMI_OK;
}
template int mfma_peak<double>(gpmi_ctx*, double*);
template int mfma_peak<float>(gpmi_ctx*, double*);

}  // namespace gpmi
