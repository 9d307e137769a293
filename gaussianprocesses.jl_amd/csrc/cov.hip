// cov.hip — covariance-matrix assembly (replaces cov!/cov_ij/distij of the reference:
// src/kernels/kernels.jl:39-71, src/kernels/stationary.jl:25-27, src/kernels/distance.jl:41-106,
// and the leaf cov(k,r) functions listed in include/gpmi.h).
//
// Design (gfx950): HBM-write bound — 8 B (fp64) per entry against ~(3d + 45) fp64 VALU ops.
//   * one 256-thread workgroup = 64 rows x (64 lanes * VEC) columns, VEC = 16 B / sizeof(T):
//     every wavefront store instruction writes 1 KiB of ONE output row (fully coalesced);
//   * the d x tile input blocks are staged in LDS with coalesced loads of x: the row block
//     as [row][k] (read back as wave-uniform broadcasts), the column block transposed
//     [k][col] so a lane's VEC columns are one conflict-free 16-B read, held in registers
//     for the whole tile (d <= 16) — no per-entry LDS traffic;
//   * the kernel tree is a postfix program interpreted with wave-uniform control flow;
//     (x-y)^2 per input row is computed once per entry and shared by all leaves;
//   * nugget, identity padding and the lower-triangle tile skip are fused in.
#include "common.h"

namespace gpmi {

namespace {

constexpr int STK = 6;  // evaluation-stack depth (validated on the host)

template <typename T>
struct Tr;
// exp(x) for x <= 0 in fp64 without the library routine's special-case handling (~19 instead of ~28 VALU instructions per
// entry of a kernel that is VALU-bound on it): Cody-Waite reduction x = k ln2 + r, |r| <= ln2 / 2, degree-13 polynomial,
// v_ldexp_f64 (which flushes the underflow to 0 by itself).  |rel. error| < 3e-16 on [-745, 0] (tests/test_gpu_parity.py pins
// cov! at rtol 1e-12 against the oracle's libm exp).
__device__ __forceinline__ double exp_nonpos(double x) {
    x = (x < -750.0) ? -750.0 : x;  // (not fmax: a NaN input must stay NaN and fail in the factorisation, as in the reference)
    const double kf = rint(x * 1.4426950408889634074);
    double r = fma(kf, -6.93147180369123816490e-01, x);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;            // 1/13!
    p = fma(p, r, 2.08767569878681e-09);          // 1/12!
    p = fma(p, r, 2.505210838544172e-08);         // 1/11!
    p = fma(p, r, 2.755731922398589e-07);         // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);        // 1/9!
    p = fma(p, r, 2.48015873015873e-05);          // 1/8!
    p = fma(p, r, 1.984126984126984e-04);         // 1/7!
    p = fma(p, r, 1.388888888888889e-03);         // 1/6!
    p = fma(p, r, 8.333333333333333e-03);         // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);        // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}

template <>
struct Tr<double> {
    static __device__ __forceinline__ double exp_(double x) { return exp_nonpos(x); }  // every leaf calls exp on -(something >= 0)
    static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
    static __device__ __forceinline__ double pow_(double x, double y) { return pow(x, y); }
    static __device__ __forceinline__ double fma_(double a, double b, double c) { return fma(a, b, c); }
    static constexpr double isapprox_rtol = 1.4901161193847656e-08;  // sqrt(eps(Float64))
};
template <>
struct Tr<float> {
    static __device__ __forceinline__ float exp_(float x) { return expf(x); }
    static __device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float pow_(float x, float y) { return powf(x, y); }
    static __device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
    static constexpr float isapprox_rtol = 3.4526698300124393e-04f;  // sqrt(eps(Float32))
};

// leaf cov(k, r): r is the (weighted) SQUARED distance; Matern leaves take its root
// (distij(::Euclidean) = sqrt, exact 0 on coincident points because r is exactly 0 there).
template <typename T>
__device__ __forceinline__ T leaf_value(int op, T r, T s2, T p0inv, T p1) {
    switch (op) {
        case GPMI_K_SE_ISO: return s2 * Tr<T>::exp_((T(-0.5) * r) * p0inv);            // se_iso.jl:39
        case GPMI_K_SE_ARD: return s2 * Tr<T>::exp_(T(-0.5) * r);                       // se_ard.jl:43
        case GPMI_K_MAT12_ISO: return s2 * Tr<T>::exp_(-(Tr<T>::sqrt_(r) * p0inv));     // mat12_iso.jl:41
        case GPMI_K_MAT12_ARD: return s2 * Tr<T>::exp_(-Tr<T>::sqrt_(r));               // mat12_ard.jl:43
        case GPMI_K_MAT32_ISO:
        case GPMI_K_MAT32_ARD: {                                                         // mat32_*.jl
            T s = T(1.7320508075688772935) * Tr<T>::sqrt_(r) * p0inv;
            return s2 * (T(1) + s) * Tr<T>::exp_(-s);
        }
        case GPMI_K_MAT52_ISO:
        case GPMI_K_MAT52_ARD: {                                                         // mat52_*.jl
            T s = T(2.2360679774997896964) * Tr<T>::sqrt_(r) * p0inv;
            return s2 * (T(1) + s + s * s * T(1.0 / 3.0)) * Tr<T>::exp_(-s);
        }
        case GPMI_K_RQ_ISO:                                                              // rq_iso.jl:44
        case GPMI_K_RQ_ARD:                                                              // rq_ard.jl:47
            return s2 * Tr<T>::pow_(T(1) + r * p0inv, -p1);
        default: return s2;  // GPMI_K_CONST (const.jl:36)
    }
}

// the same without the RQ leaves (fp64 pow drags ~90 VGPRs into whatever kernel contains it)
template <typename T>
__device__ __forceinline__ T leaf_value_nopow(int op, T r, T s2, T p0inv) {
    switch (op) {
        case GPMI_K_SE_ISO: return s2 * Tr<T>::exp_((T(-0.5) * r) * p0inv);
        case GPMI_K_SE_ARD: return s2 * Tr<T>::exp_(T(-0.5) * r);
        case GPMI_K_MAT12_ISO: return s2 * Tr<T>::exp_(-(Tr<T>::sqrt_(r) * p0inv));
        case GPMI_K_MAT12_ARD: return s2 * Tr<T>::exp_(-Tr<T>::sqrt_(r));
        case GPMI_K_MAT32_ISO:
        case GPMI_K_MAT32_ARD: {
            T s = T(1.7320508075688772935) * Tr<T>::sqrt_(r) * p0inv;
            return s2 * (T(1) + s) * Tr<T>::exp_(-s);
        }
        case GPMI_K_MAT52_ISO:
        case GPMI_K_MAT52_ARD: {
            T s = T(2.2360679774997896964) * Tr<T>::sqrt_(r) * p0inv;
            return s2 * (T(1) + s + s * s * T(1.0 / 3.0)) * Tr<T>::exp_(-s);
        }
        default: return s2;
    }
}

// Is this tile one the fast kernel takes?  One stationary leaf (the bench's SEArd; any single SE / Matern / RQ kernel) and a
// tile that needs neither padding nor the diagonal (nugget).  Both kernels evaluate it, so every tile has exactly one owner.
__device__ __forceinline__ bool fast_tile(const DevProgram* __restrict__ prog, int nops, int flags, int64_t row0, int64_t col0,
                                          int TR, int TC, int64_t na, int64_t nb, int64_t nrows, int64_t ncols, int64_t row_off) {
    if (nops != 1) {
        if (prog->fast_class < 0) return false;  // multi-leaf: the specialised kernel's programs only
    } else {
        const int op0 = prog->leaf[0].op;
        if (op0 == GPMI_K_NOISE || op0 == GPMI_K_CONST) return false;
    }
    const bool interior = row0 + TR <= na && row0 + TR <= nrows && col0 + TC <= nb && col0 + TC <= ncols;
    const bool on_diag = (flags & COV_NUGGET) && col0 <= row_off + row0 + TR - 1 && row_off + row0 <= col0 + TC - 1;
    return interior && !on_diag;
}

// cov_fast_kernel: the same tiling as cov_kernel for the tiles fast_tile() selects.  Weights and leaf constants are read
// once, there is no evaluation stack and no per-entry edge logic: about half the VALU instructions per entry of the
// interpreter, and few enough registers for 6+ waves per SIMD (the interpreter needs 177 VGPRs: 2 waves).
template <typename T, int DMAX>
__global__ __launch_bounds__(256) void cov_fast_kernel(const T* __restrict__ xa, int64_t na, const T* __restrict__ xb,
                                                       int64_t nb, int d, T* __restrict__ C, int64_t ldc, int64_t nrows,
                                                       int64_t ncols, const DevProgram* __restrict__ prog, int flags,
                                                       int64_t row_off) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int TC = 64 * VEC;
    constexpr int TR = 64;
    using VT = T __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sa = reinterpret_cast<T*>(smem);  // [TR][d]
    T* sbT = sa + TR * d;                // [d][TC]
    const int64_t row0 = (int64_t)blockIdx.y * TR;
    const int64_t col0 = (int64_t)blockIdx.x * TC;
    if ((flags & COV_LOWER) && col0 > row_off + row0 + TR - 1) return;  // tile strictly above the diagonal
    if (!fast_tile(prog, prog->n_ops, flags, row0, col0, TR, TC, na, nb, nrows, ncols, row_off)) return;
    const int tid = threadIdx.x;
    for (int e = tid; e < TR * d; e += 256) sa[e] = xa[(row0 + e / d) * d + (e % d)];
    for (int e = tid; e < TC * d; e += 256) {
        const int c = e / d, k = e - c * d;
        sbT[k * TC + c] = xb[(col0 + c) * d + k];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    T xbr[DMAX][VEC];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        if (k < d) {
            const VT v = *reinterpret_cast<const VT*>(&sbT[k * TC + lane * VEC]);
#pragma unroll
            for (int q = 0; q < VEC; ++q) xbr[k][q] = v[q];
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) xbr[k][q] = T(0);
        }
    }
    const int op0 = prog->leaf[0].op;
    const double* w = prog->w + prog->leaf[0].woff;
    T wk[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) wk[k] = k < d ? (T)w[k] : T(0);
    const T s2 = (T)prog->leaf[0].s2, p0inv = (T)prog->leaf[0].p0, p1 = (T)prog->leaf[0].p1;
    for (int rr = 0; rr < TR / 4; ++rr) {
        const int row = wave * (TR / 4) + rr;
        const T* sar = sa + row * d;
        T r[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) r[q] = T(0);
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            if (k < d) {
                const T a = sar[k];
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const T df = a - xbr[k][q];
                    r[q] = Tr<T>::fma_(df * df, wk[k], r[q]);  // same operation order as the interpreter
                }
            }
        }
        VT out;
#pragma unroll
        for (int q = 0; q < VEC; ++q) out[q] = leaf_value<T>(op0, r[q], s2, p0inv, p1);
        *reinterpret_cast<VT*>(&C[(row0 + row) * ldc + col0 + (int64_t)lane * VEC]) = out;
    }
}

// cov_multi_kernel: the interior, off-diagonal tiles of a MULTI-LEAF program (Sum / Prod of stationary, Const and Noise leaves,
// evaluation depth <= 3: DevProgram::fast_class) — BASELINE configs[2], (SEArd + Mat52Iso) + Noise, is one.  Same tiling as
// cov_fast_kernel; against the generic interpreter it has no per-entry edge logic, a three-deep stack instead of a six-deep
// one, no fp64 pow unless the program has an RQ leaf (FEAT & 1: the library routine alone costs ~90 VGPRs) and the Noise leaf
// (FEAT & 2) behind a prefilter: noise.jl:31-37 asks x_z ~ y_z for every active row z (isapprox, rtol sqrt(eps)), which needs
// (x_z - y_z)^2 <= (rtol max|x|)^2 for all z — one v_max per row on the squared differences the other leaves need anyway; the
// exact test runs only where some lane of the wave passes it (coincident points: the diagonal, duplicates).
template <typename T, int DMAX, int FEAT>
__global__ __launch_bounds__(256) void cov_multi_kernel(const T* __restrict__ xa, int64_t na, const T* __restrict__ xb,
                                                        int64_t nb, int d, T* __restrict__ C, int64_t ldc, int64_t nrows,
                                                        int64_t ncols, const DevProgram* __restrict__ prog, int flags,
                                                        int64_t row_off) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int TC = 64 * VEC;
    constexpr int TR = 64;
    using VT = T __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sa = reinterpret_cast<T*>(smem);  // [TR][d]
    T* sbT = sa + TR * d;                // [d][TC]
    __shared__ T s_max[4];
    const int64_t row0 = (int64_t)blockIdx.y * TR;
    const int64_t col0 = (int64_t)blockIdx.x * TC;
    if ((flags & COV_LOWER) && col0 > row_off + row0 + TR - 1) return;  // tile strictly above the diagonal
    const int nops = prog->n_ops;
    if (nops == 1 || !fast_tile(prog, nops, flags, row0, col0, TR, TC, na, nb, nrows, ncols, row_off)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    T amax = T(0);
    for (int e = tid; e < TR * d; e += 256) {
        const T v = xa[(row0 + e / d) * d + (e % d)];
        sa[e] = v;
        if constexpr (FEAT & 2) amax = fabs(v) > amax ? fabs(v) : amax;
    }
    for (int e = tid; e < TC * d; e += 256) {
        const int c = e / d, k = e - c * d;
        const T v = xb[(col0 + c) * d + k];
        sbT[k * TC + c] = v;
        if constexpr (FEAT & 2) amax = fabs(v) > amax ? fabs(v) : amax;
    }
    if constexpr (FEAT & 2) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const T o = __shfl_xor(amax, off, 64);
            amax = o > amax ? o : amax;
        }
        if (lane == 0) s_max[wave] = amax;
    }
    __syncthreads();
    T thr2 = T(0);
    if constexpr (FEAT & 2) {
        T m = s_max[0];
        for (int w = 1; w < 4; ++w) m = s_max[w] > m ? s_max[w] : m;
        const T thr = Tr<T>::isapprox_rtol * m * T(1.0000001);
        thr2 = thr * thr;
    }
    T xbr[DMAX][VEC];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
        if (k < d) {
            const VT v = *reinterpret_cast<const VT*>(&sbT[k * TC + lane * VEC]);
#pragma unroll
            for (int q = 0; q < VEC; ++q) xbr[k][q] = v[q];
        } else {
#pragma unroll
            for (int q = 0; q < VEC; ++q) xbr[k][q] = T(0);
        }
    }
    for (int rr = 0; rr < TR / 4; ++rr) {
        const int row = wave * (TR / 4) + rr;
        const T* sar = sa + row * d;
        T dsq[DMAX][VEC];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            const T a = k < d ? sar[k] : T(0);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const T df = a - xbr[k][q];
                dsq[k][q] = df * df;
            }
        }
        T s0[VEC], s1[VEC], s2[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) s0[q] = s1[q] = s2[q] = T(0);
        for (int o = 0; o < nops; ++o) {  // wave-uniform control flow throughout
            const int op = prog->leaf[o].op;
            if (op == GPMI_K_SUM || op == GPMI_K_PROD) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    s0[q] = (op == GPMI_K_SUM) ? (s1[q] + s0[q]) : (s1[q] * s0[q]);
                    s1[q] = s2[q];
                }
                continue;
            }
            T val[VEC];
            const T sig2 = (T)prog->leaf[o].s2;
            const double* w = prog->w + prog->leaf[o].woff;
            if (op == GPMI_K_CONST) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) val[q] = sig2;
            } else if (op == GPMI_K_NOISE) {
                if constexpr (FEAT & 2) {
                    T mx[VEC];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) mx[q] = T(0);
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) {
                        if (k < d && w[k] != 0.0) {
#pragma unroll
                            for (int q = 0; q < VEC; ++q) mx[q] = dsq[k][q] > mx[q] ? dsq[k][q] : mx[q];
                        }
                    }
                    bool cand = false;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) cand = cand || (mx[q] <= thr2);
#pragma unroll
                    for (int q = 0; q < VEC; ++q) val[q] = T(0);
                    if (__any(cand)) {  // the exact test of noise.jl:31-37, as the interpreter does it
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            bool same = true;
                            for (int k = 0; k < d; ++k) {
                                if (w[k] != 0.0) {
                                    const T a = sar[k], b = sbT[k * TC + lane * VEC + q];
                                    const T m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
                                    same = same && ((a == b) || (fabs(a - b) <= Tr<T>::isapprox_rtol * m));
                                }
                            }
                            val[q] = same ? sig2 : T(0);
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) val[q] = T(0);
                }
            } else {
                T r[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) r[q] = T(0);
#pragma unroll
                for (int k = 0; k < DMAX; ++k) {
                    if (k < d) {
                        const T wk = (T)w[k];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) r[q] = Tr<T>::fma_(dsq[k][q], wk, r[q]);
                    }
                }
                const T p0inv = (T)prog->leaf[o].p0, p1 = (T)prog->leaf[o].p1;
                if ((FEAT & 1) && (op == GPMI_K_RQ_ISO || op == GPMI_K_RQ_ARD)) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) val[q] = sig2 * Tr<T>::pow_(T(1) + r[q] * p0inv, -p1);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) val[q] = leaf_value_nopow<T>(op, r[q], sig2, p0inv);
                }
            }
#pragma unroll
            for (int q = 0; q < VEC; ++q) {  // push
                s2[q] = s1[q];
                s1[q] = s0[q];
                s0[q] = val[q];
            }
        }
        VT out;
#pragma unroll
        for (int q = 0; q < VEC; ++q) out[q] = s0[q];
        *reinterpret_cast<VT*>(&C[(row0 + row) * ldc + col0 + (int64_t)lane * VEC]) = out;
    }
}

template <typename T, int DMAX>
__global__ __launch_bounds__(256) void cov_kernel(const T* __restrict__ xa, int64_t na, const T* __restrict__ xb,
                                                  int64_t nb, int d, T* __restrict__ C, int64_t ldc, int64_t nrows,
                                                  int64_t ncols, const DevProgram* __restrict__ prog, int flags,
                                                  double nugget, const double* __restrict__ nugget_vec,
                                                  int64_t row_off) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int TC = 64 * VEC;
    constexpr int TR = 64;
    using VT = T __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sa = reinterpret_cast<T*>(smem);  // [TR][d]
    T* sbT = sa + TR * d;                // [d][TC]

    const int64_t row0 = (int64_t)blockIdx.y * TR;
    const int64_t col0 = (int64_t)blockIdx.x * TC;
    // row_off: global index of local row 0 (a shard assembles only its own block-rows of K)
    if ((flags & COV_LOWER) && col0 > row_off + row0 + TR - 1) return;  // tile strictly above the diagonal

    const int tid = threadIdx.x;
    for (int e = tid; e < TR * d; e += 256) {
        int r = e / d, k = e - r * d;
        int64_t gr = row0 + r;
        gr = gr < na ? gr : na - 1;
        gr = gr < 0 ? 0 : gr;
        sa[e] = xa[gr * d + k];
    }
    for (int e = tid; e < TC * d; e += 256) {
        int c = e / d, k = e - c * d;
        int64_t gc = col0 + c;
        gc = gc < nb ? gc : nb - 1;
        sbT[k * TC + c] = xb[gc * d + k];
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nops = prog->n_ops;
    const bool has_noise = prog->has_noise_leaf != 0;

    T xbr[DMAX > 0 ? DMAX : 1][VEC];
    if constexpr (DMAX > 0) {
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            if (k < d) {
                VT v = *reinterpret_cast<const VT*>(&sbT[k * TC + lane * VEC]);
#pragma unroll
                for (int q = 0; q < VEC; ++q) xbr[k][q] = v[q];
            } else {
#pragma unroll
                for (int q = 0; q < VEC; ++q) xbr[k][q] = T(0);
            }
        }
    }

    // interior, off-diagonal tiles of a single stationary leaf belong to cov_fast_kernel (launched alongside)
    if constexpr (DMAX > 0) {
        if (fast_tile(prog, nops, flags, row0, col0, TR, TC, na, nb, nrows, ncols, row_off)) return;
    }

    for (int rr = 0; rr < TR / 4; ++rr) {
        const int row = wave * (TR / 4) + rr;
        const int64_t grow = row0 + row;
        if (grow >= nrows) break;
        const T* sar = sa + row * d;

        T dsq[DMAX > 0 ? DMAX : 1][VEC];
        if constexpr (DMAX > 0) {
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {
                if (k < d) {
                    T a = sar[k];
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        T df = a - xbr[k][q];
                        dsq[k][q] = df * df;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) dsq[k][q] = T(0);
                }
            }
        }

        T st[STK][VEC];
#pragma unroll
        for (int s = 0; s < STK; ++s)
#pragma unroll
            for (int q = 0; q < VEC; ++q) st[s][q] = T(0);

        for (int o = 0; o < nops; ++o) {
            const int op = prog->leaf[o].op;
            if (op == GPMI_K_SUM || op == GPMI_K_PROD) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) st[0][q] = (op == GPMI_K_SUM) ? (st[1][q] + st[0][q]) : (st[1][q] * st[0][q]);
#pragma unroll
                for (int s = 1; s < STK - 1; ++s)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) st[s][q] = st[s + 1][q];
                continue;
            }
            T val[VEC];
            const T s2 = (T)prog->leaf[o].s2;
            if (op == GPMI_K_CONST) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) val[q] = s2;
            } else if (op == GPMI_K_NOISE) {
                // noise.jl:31-37: all active rows z satisfy X1[z,i] ≈ X2[z,j]
                const double* w = prog->w + prog->leaf[o].woff;
                bool same[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) same[q] = true;
                for (int k = 0; k < d; ++k) {
                    if (w[k] != 0.0) {
                        T a = sar[k];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            T b = sbT[k * TC + lane * VEC + q];
                            T m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
                            bool ok = (a == b) || (fabs(a - b) <= Tr<T>::isapprox_rtol * m);
                            same[q] = same[q] && ok;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < VEC; ++q) val[q] = same[q] ? s2 : T(0);
            } else {
                const double* w = prog->w + prog->leaf[o].woff;
                T r[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) r[q] = T(0);
                if constexpr (DMAX > 0) {
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) {
                        if (k < d) {
                            T wk = (T)w[k];
#pragma unroll
                            for (int q = 0; q < VEC; ++q) r[q] = Tr<T>::fma_(dsq[k][q], wk, r[q]);
                        }
                    }
                } else {
                    for (int k = 0; k < d; ++k) {
                        T wk = (T)w[k];
                        T a = sar[k];
                        VT bv = *reinterpret_cast<const VT*>(&sbT[k * TC + lane * VEC]);
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            T df = a - bv[q];
                            r[q] = Tr<T>::fma_(df * df, wk, r[q]);
                        }
                    }
                }
                const T p0inv = (T)prog->leaf[o].p0;
                const T p1 = (T)prog->leaf[o].p1;
#pragma unroll
                for (int q = 0; q < VEC; ++q) val[q] = leaf_value<T>(op, r[q], s2, p0inv, p1);
            }
            // push
#pragma unroll
            for (int s = STK - 1; s > 0; --s)
#pragma unroll
                for (int q = 0; q < VEC; ++q) st[s][q] = st[s - 1][q];
#pragma unroll
            for (int q = 0; q < VEC; ++q) st[0][q] = val[q];
        }
        (void)has_noise;

        const int64_t gcol0 = col0 + (int64_t)lane * VEC;
        VT out;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int64_t gcol = gcol0 + q;
            T v = st[0][q];
            const int64_t gg = row_off + grow;  // global row
            if (grow >= na || gcol >= nb) {
                v = ((flags & COV_PAD_IDENTITY) && gg == gcol) ? T(1) : T(0);
            } else if ((flags & COV_NUGGET) && gg == gcol) {
                v += nugget_vec ? (T)nugget_vec[gg] : (T)nugget;  // GPE.jl:173,181-183 / GP.jl:104-108
            }
            out[q] = v;
        }
        if (gcol0 + VEC <= ncols) *reinterpret_cast<VT*>(&C[grow * ldc + gcol0]) = out;
    }
}

template <typename T, int DMAX>
void launch_cov_t(gpmi_ctx* ctx, const T* xa, int64_t na, const T* xb, int64_t nb, int d, T* C, int64_t ldc,
                  int64_t nrows_total, int64_t ncols_total, int flags, double nugget, const double* nugget_vec,
                  int64_t row_off) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int TC = 64 * VEC;
    constexpr int TR = 64;
    dim3 grid((unsigned)((ncols_total + TC - 1) / TC), (unsigned)((nrows_total + TR - 1) / TR));
    size_t lds = (size_t)(TR + TC) * d * sizeof(T);
    auto kern = cov_kernel<T, DMAX>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, xa, na, xb, nb, d, C, ldc, nrows_total, ncols_total,
                       ctx->d_prog, flags, nugget, nugget_vec, row_off);
    if constexpr (DMAX > 0) {
        const DevProgram* hp = ctx->h_prog;
        if (hp->n_ops == 1 && hp->leaf[0].op != GPMI_K_NOISE && hp->leaf[0].op != GPMI_K_CONST) {
            auto fk = cov_fast_kernel<T, DMAX>;
            hipLaunchKernelGGL(fk, grid, dim3(256), lds, ctx->stream, xa, na, xb, nb, d, C, ldc, nrows_total, ncols_total,
                               ctx->d_prog, flags, row_off);
        } else if (hp->n_ops > 1 && hp->fast_class >= 0) {
            auto go = [&](auto mk) {
                hipLaunchKernelGGL(mk, grid, dim3(256), lds, ctx->stream, xa, na, xb, nb, d, C, ldc, nrows_total, ncols_total, ctx->d_prog, flags,
                                   row_off);
            };
            switch (hp->fast_class) {
                case 0: go(cov_multi_kernel<T, DMAX, 0>); break;
                case 1: go(cov_multi_kernel<T, DMAX, 1>); break;
                case 2: go(cov_multi_kernel<T, DMAX, 2>); break;
                default: go(cov_multi_kernel<T, DMAX, 3>); break;
            }
        }
    }
}

}  // namespace

template <typename T>
void launch_cov(gpmi_ctx* ctx, const T* xa, int64_t na, const T* xb, int64_t nb, int d, T* C, int64_t ldc,
                int64_t nrows_total, int64_t ncols_total, int flags, double nugget, const double* nugget_vec,
                int64_t row_off) {
    // algorithmic bytes: one write per generated entry (lower-triangle tiles only when COV_LOWER)
    double entries = (flags & COV_LOWER) ? 0.5 * (double)nrows_total * ((double)ncols_total + 1.0)
                                         : (double)nrows_total * (double)ncols_total;
    ProfScope ps(ctx, GPMI_PROF_COV, entries * sizeof(T));
    if (d <= 4)
        launch_cov_t<T, 4>(ctx, xa, na, xb, nb, d, C, ldc, nrows_total, ncols_total, flags, nugget, nugget_vec, row_off);
    else if (d <= 8)
        launch_cov_t<T, 8>(ctx, xa, na, xb, nb, d, C, ldc, nrows_total, ncols_total, flags, nugget, nugget_vec, row_off);
    else if (d <= 16)
        launch_cov_t<T, 16>(ctx, xa, na, xb, nb, d, C, ldc, nrows_total, ncols_total, flags, nugget, nugget_vec, row_off);
    else
        launch_cov_t<T, 0>(ctx, xa, na, xb, nb, d, C, ldc, nrows_total, ncols_total, flags, nugget, nugget_vec, row_off);
}

template void launch_cov<double>(gpmi_ctx*, const double*, int64_t, const double*, int64_t, int, double*, int64_t,
                                 int64_t, int64_t, int, double, const double*, int64_t);
template void launch_cov<float>(gpmi_ctx*, const float*, int64_t, const float*, int64_t, int, float*, int64_t, int64_t,
                                int64_t, int, double, const double*, int64_t);

}  // namespace gpmi
