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

// Is this tile one the fast kernel takes?  One stationary leaf (the bench's SEArd; any single SE / Matern / RQ kernel) and a
// tile that needs neither padding nor the diagonal (nugget).  Both kernels evaluate it, so every tile has exactly one owner.
__device__ __forceinline__ bool fast_tile(const DevProgram* __restrict__ prog, int nops, int flags, int64_t row0, int64_t col0,
                                          int TR, int TC, int64_t na, int64_t nb, int64_t nrows, int64_t ncols, int64_t row_off) {
    if (flags & COV_NO_FAST) return false;
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

// ---- single stationary leaf: cov_leaf_kernel (round 4; replaces round 1's cov_fast_kernel) ------------------------------------
// The PMC record of the old kernel (profiles/r04_a_cov_pmc_before.json) says what bounded it: 70 VALU lane-instructions per entry
// (fp64, d = 8) at 67 % VALU issue utilisation — twice what the arithmetic needs, because the leaf switch, the `k < d` tests and the
// polynomial's literal constants (re-materialised by v_mov_b64) all sat inside the row loop.  This kernel does per entry only
//   d x (v_add_f64, v_fma_f64)   the squared distance of PRE-SCALED inputs: x_k sqrt(w_k c) is formed once per call by
//                                scale_inputs_kernel into zero-padded rows of DMAX (N d elements against N^2 entries), so the ARD /
//                                iso / Masked weight and the leaf's reciprocal length constant cost nothing here;
//   the leaf function            on the family FAM (template: no switch), exp by exp_nonpos with its constants in scalar registers;
// the row operand comes through SCALAR loads (the row pointer is wave-uniform: s_load_dwordx8/16, no LDS, no barrier), the column
// operand is loaded once per wave into registers (this lane's VEC columns: DMAX contiguous elements each), two rows per iteration
// for instruction-level parallelism.  Same tiles, same owner rule (fast_tile) and the same 1 KiB-per-row wavefront stores as before.
enum LeafFamily { FAM_SE = 0, FAM_MAT12 = 1, FAM_MAT32 = 2, FAM_MAT52 = 3, FAM_RQ = 4 };

template <int DMAX>
struct ScaleW {
    double sw[DMAX];  // sqrt(w_k * c): c = the leaf's reciprocal length constant folded into the weights (host, fp64)
};

// amax (may be null): running maximum of |x| over everything scaled so far, as the bit pattern of a non-negative IEEE number
// (ordered like the number itself) — the Noise prefilter's scale
// shift (may be null): ONE point subtracted from every row before scaling — the same point for both blocks of a call.  The reference's
// distance loops difference the raw inputs (distance.jl:41-106: (x_k - y_k)^2 w_k), whose rounding error is relative to |x - y|; scaling
// first would make it relative to |x| (inputs with a large common offset — years, timestamps — lose digits in r).  With both blocks
// centred on a common data point the rounding of (x_k - c_k) is relative to the data's SPREAD, and (x_k - c_k) - (y_k - c_k) = x_k - y_k.
// The centre is PER CALL (the first point of the call's column block xb).  Every assembly of K for a factorisation — gpmi_fit and the
// blocked handle's assemble(), whatever the rank or stripe — passes the WHOLE training set as xb, so all of K shares one centre (the first
// training point) and an entry has the same bits whichever call or rank generated it.  The cross-covariance blocks of predict (blocked:
// one call per column block, centred on that block's first point) are independent of each other: entries of DIFFERENT calls agree only to
// rounding, and nothing downstream assumes more.
template <typename T, int DMAX>
__global__ __launch_bounds__(256) void scale_inputs_kernel(const T* __restrict__ x, int64_t n, int d, ScaleW<DMAX> w, T* __restrict__ out,
                                                           unsigned long long* __restrict__ amax, const T* __restrict__ shift) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    T av = T(0);
    if (i < n * DMAX) {
        const int64_t row = i / DMAX;
        const int k = (int)(i - row * DMAX);
        const T v = k < d ? x[row * d + k] : T(0);
        av = fabs(v);
        const T c = (shift && k < d) ? shift[k] : T(0);
        out[i] = (T)((double)(v - c) * w.sw[k]);
    }
    if (amax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const T o = __shfl_xor(av, off, 64);
            av = o > av ? o : av;
        }
        if ((threadIdx.x & 63) == 0) {
            unsigned long long bits;
            if constexpr (sizeof(T) == 8)
                bits = (unsigned long long)__double_as_longlong((double)av);
            else
                bits = (unsigned long long)__float_as_uint((float)av);
            if (bits > *amax) atomicMax(amax, bits);  // (6250 waves on one word otherwise: 73 us per call)
        }
    }
}

// sqrt(r) for r >= 0 in fp64 without the library's scaling / special cases: v_rsq_f64 + two Goldschmidt steps (|rel. error| < 2e-16);
// r = 0 (coincident points) gives exactly 0, as distij(::Euclidean) does
__device__ __forceinline__ double sqrt_nonneg(double r) {
    const double rc = fmax(r, 1e-300);
    double y = __builtin_amdgcn_rsq(rc);
    double g = rc * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double res = fma(fma(-g, g, rc), h, g);
    return r > 0.0 ? res : 0.0;
}
__device__ __forceinline__ float sqrt_nonneg(float r) { return sqrtf(r); }

// the leaf function on the squared distance of the PRE-SCALED inputs (file header of each kernel in include/gpmi.h)
template <typename T, int FAM>
__device__ __forceinline__ T leaf_family(T r, T s2, T p1) {
    if constexpr (FAM == FAM_SE) {
        return s2 * Tr<T>::exp_(T(-0.5) * r);                                   // se_iso.jl:39, se_ard.jl:43 (1/l2 in the weights)
    } else if constexpr (FAM == FAM_MAT12) {
        return s2 * Tr<T>::exp_(-sqrt_nonneg(r));                               // mat12_*.jl (1/l^2 in the weights)
    } else if constexpr (FAM == FAM_MAT32) {
        const T s = sqrt_nonneg(r);                                             // mat32_*.jl (3/l^2 in the weights)
        return s2 * (T(1) + s) * Tr<T>::exp_(-s);
    } else if constexpr (FAM == FAM_MAT52) {
        const T s = sqrt_nonneg(r);                                             // mat52_*.jl (5/l^2 in the weights): 1 + s + s^2/3
        return s2 * Tr<T>::fma_(s, Tr<T>::fma_(s, T(1.0 / 3.0), T(1)), T(1)) * Tr<T>::exp_(-s);
    } else {
        return s2 * Tr<T>::pow_(T(1) + r, -p1);                                 // rq_*.jl (1/(2 a l2) resp. 0.5/a in the weights)
    }
}

template <typename T, int DMAX, int FAM>
__global__ __launch_bounds__(256) void cov_leaf_kernel(const T* __restrict__ xas, int64_t na, const T* __restrict__ xbs, int64_t nb,
                                                       T* __restrict__ C, int64_t ldc, int64_t nrows, int64_t ncols,
                                                       const DevProgram* __restrict__ prog, int flags, int64_t row_off, T s2, T p1) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int TC = 64 * VEC;
    constexpr int TR = 64;
    using VT = T __attribute__((ext_vector_type(VEC)));
    const int64_t row0 = (int64_t)blockIdx.y * TR;
    const int64_t col0 = (int64_t)blockIdx.x * TC;
    if ((flags & COV_LOWER) && col0 > row_off + row0 + TR - 1) return;  // tile strictly above the diagonal
    if (!fast_tile(prog, 1, flags, row0, col0, TR, TC, na, nb, nrows, ncols, row_off)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // provably wave-uniform: the row operand goes through SMEM
    // this lane's VEC columns of the (scaled, zero-padded) column block: DMAX contiguous elements each
    T xbr[DMAX][VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        const T* pb = xbs + (col0 + (int64_t)lane * VEC + q) * DMAX;
#pragma unroll
        for (int k = 0; k < DMAX; k += VEC) {
            const VT v = *reinterpret_cast<const VT*>(pb + k);
#pragma unroll
            for (int j = 0; j < VEC; ++j) xbr[k + j][q] = v[j];
        }
    }
    // (Measured and not kept, GPU call G: the row operand in row PAIRS so that fp32 runs the distance loop as v_pk_add_f32 / v_pk_fma_f32 over
    //  two rows — 16 instead of 24 VALU instructions per entry, but the pre-splatted column registers double (164 VGPRs at d = 16,
    //  3 waves per SIMD instead of 6): 1.59 against 1.40 ms in fp32, 2.09 against 1.98 ms in fp64.)
    const T* __restrict__ ar = xas + (row0 + wave * (TR / 4)) * DMAX;
    T* __restrict__ crow = C + (row0 + wave * (TR / 4)) * ldc + col0 + (int64_t)lane * VEC;
#pragma unroll 1
    for (int rr = 0; rr < TR / 4; rr += 2) {
        T r0[VEC], r1[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) r0[q] = r1[q] = T(0);
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            const T a0 = ar[rr * DMAX + k], a1 = ar[(rr + 1) * DMAX + k];  // wave-uniform addresses: scalar loads
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const T d0 = a0 - xbr[k][q], d1 = a1 - xbr[k][q];
                r0[q] = Tr<T>::fma_(d0, d0, r0[q]);
                r1[q] = Tr<T>::fma_(d1, d1, r1[q]);
            }
        }
        VT o0, o1;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            o0[q] = leaf_family<T, FAM>(r0[q], s2, p1);
            o1[q] = leaf_family<T, FAM>(r1[q], s2, p1);
        }
        *reinterpret_cast<VT*>(crow + (int64_t)rr * ldc) = o0;
        *reinterpret_cast<VT*>(crow + (int64_t)(rr + 1) * ldc) = o1;
    }
}

// cov_multi_kernel: the interior, off-diagonal tiles of a MULTI-LEAF program (Sum / Prod of stationary, Const and Noise leaves,
// evaluation depth <= 3: DevProgram::fast_class) — BASELINE configs[2], (SEArd + Mat52Iso) + Noise, is one.  Round 4 rewrite on the
// PMC record of round 3's form (profiles/r04_a_cov_pmc_before.json: 204 VALU lane-instructions per entry at 75 % VALU issue utilisation — the
// arithmetic needs ~110; the rest were `k < d` selects, literal-constant moves and per-row scalar loads of the leaf parameters with
// their waits): the operands come zero-padded to DMAX (scale_inputs_kernel with unit weights: no k < d tests; the row operand
// through scalar loads, the column operand straight into registers — no LDS transposition, no barrier after the prologue), every
// leaf's parameters and weights sit in LDS in the element type (one broadcast read per use instead of an SMEM round trip), TWO rows
// per pass over the program (the wave-uniform control flow is paid once per four / eight entries of a lane), sqrt by rsq + Goldschmidt.
// No fp64 pow unless the program has an RQ leaf (FEAT & 1: the library routine alone costs ~90 VGPRs); the Noise leaf (FEAT & 2)
// behind a prefilter: noise.jl:31-37 asks x_z ~ y_z for every active row z (isapprox, rtol sqrt(eps)), which needs
// (x_z - y_z)^2 <= (rtol max|x|)^2 for all z — one v_max per row on the squared differences the other leaves need anyway; the exact
// test runs only where some lane of the wave passes it (coincident points: the diagonal, duplicates).
// one-instruction maximum (fmax() canonicalises both operands first: three v_max per call)
__device__ __forceinline__ double max_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <typename T, int DMAX>
struct LeafLds {
    T w[DMAX];
    T s2, p0, p1;
    int op, actmask;  // actmask: bit k set when the leaf acts on input row k (w_k != 0)
};

template <typename T, int DMAX, int FEAT>
__global__ __launch_bounds__(256) void cov_multi_kernel(const T* __restrict__ xas, int64_t na, const T* __restrict__ xbs, int64_t nb,
                                                        T* __restrict__ C, int64_t ldc, int64_t nrows, int64_t ncols,
                                                        const DevProgram* __restrict__ prog, int flags, int64_t row_off,
                                                        const unsigned long long* __restrict__ amax) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int TC = 64 * VEC;
    constexpr int TR = 64;
    using VT = T __attribute__((ext_vector_type(VEC)));
    __shared__ LeafLds<T, DMAX> s_leaf[GPMI_MAX_OPS];
    const int64_t row0 = (int64_t)blockIdx.y * TR;
    const int64_t col0 = (int64_t)blockIdx.x * TC;
    if ((flags & COV_LOWER) && col0 > row_off + row0 + TR - 1) return;  // tile strictly above the diagonal
    const int nops = prog->n_ops;
    if (nops == 1 || !fast_tile(prog, nops, flags, row0, col0, TR, TC, na, nb, nrows, ncols, row_off)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int d = prog->d;
    for (int e = tid; e < nops * (DMAX + 5); e += 256) {  // the program, once per workgroup
        const int o = e / (DMAX + 5), j = e - o * (DMAX + 5);
        const DevLeaf& lf = prog->leaf[o];
        if (j < DMAX)
            s_leaf[o].w[j] = (j < d && lf.op < GPMI_K_SUM) ? (T)prog->wtab()[lf.woff + j] : T(0);
        else if (j == DMAX)
            s_leaf[o].s2 = (T)lf.s2;
        else if (j == DMAX + 1)
            s_leaf[o].p0 = (T)lf.p0;
        else if (j == DMAX + 2)
            s_leaf[o].p1 = (T)lf.p1;
        else if (j == DMAX + 3)
            s_leaf[o].op = lf.op;
        else {
            int m = 0;
            if (lf.op < GPMI_K_SUM)
                for (int k = 0; k < d && k < DMAX; ++k) m |= (prog->wtab()[lf.woff + k] != 0.0) ? (1 << k) : 0;
            s_leaf[o].actmask = m;
        }
    }
    // this lane's VEC columns of the zero-padded column block
    T xbr[DMAX][VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
        const T* pb = xbs + (col0 + (int64_t)lane * VEC + q) * DMAX;
#pragma unroll
        for (int k = 0; k < DMAX; k += VEC) {
            const VT v = *reinterpret_cast<const VT*>(pb + k);
#pragma unroll
            for (int j = 0; j < VEC; ++j) xbr[k + j][q] = v[j];
        }
    }
    __syncthreads();
    // Noise prefilter threshold: isapprox needs |a - b| <= rtol max(|a|, |b|) <= rtol xmax (xmax: max |x| over both input blocks)
    T xmax;
    if constexpr (sizeof(T) == 8)
        xmax = (T)__longlong_as_double((long long)amax[0]);
    else
        xmax = (T)__uint_as_float((unsigned)amax[0]);
    const T thr = Tr<T>::isapprox_rtol * xmax * T(1.0000001);
    const T thr2 = thr * thr;
    const T* __restrict__ ar = xas + (row0 + wave * (TR / 4)) * DMAX;
    T* __restrict__ crow = C + (row0 + wave * (TR / 4)) * ldc + col0 + (int64_t)lane * VEC;
    constexpr int NR = 2;  // rows per pass
#pragma unroll 1
    for (int rr = 0; rr < TR / 4; rr += NR) {
        T dsq[NR][DMAX][VEC];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const T a = ar[(rr + i) * DMAX + k];  // wave-uniform address: scalar load
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const T df = a - xbr[k][q];
                    dsq[i][k][q] = df * df;
                }
            }
        }
        T s0[NR][VEC], s1[NR][VEC], s2[NR][VEC];
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < VEC; ++q) s0[i][q] = s1[i][q] = s2[i][q] = T(0);
#pragma unroll 1
        for (int o = 0; o < nops; ++o) {  // wave-uniform control flow throughout
            const int op = __builtin_amdgcn_readfirstlane(s_leaf[o].op);
            if (op == GPMI_K_SUM || op == GPMI_K_PROD) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        s0[i][q] = (op == GPMI_K_SUM) ? (s1[i][q] + s0[i][q]) : (s1[i][q] * s0[i][q]);
                        s1[i][q] = s2[i][q];
                    }
                continue;
            }
            T val[NR][VEC];
            const T sig2 = s_leaf[o].s2;
            if (op == GPMI_K_CONST) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) val[i][q] = sig2;
            } else if (op == GPMI_K_NOISE) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) val[i][q] = T(0);
                if constexpr (FEAT & 2) {
                    T mx[NR][VEC];
#pragma unroll
                    for (int i = 0; i < NR; ++i)
#pragma unroll
                        for (int q = 0; q < VEC; ++q) mx[i][q] = T(0);
                    const int am = __builtin_amdgcn_readfirstlane(s_leaf[o].actmask);  // wave-uniform: scalar branches, one v_max per element
#pragma unroll
                    for (int k = 0; k < DMAX; ++k) {
                        if ((am >> k) & 1) {
#pragma unroll
                            for (int i = 0; i < NR; ++i)
#pragma unroll
                                for (int q = 0; q < VEC; ++q) mx[i][q] = max_raw(mx[i][q], dsq[i][k][q]);  // (a NaN difference leaves mx alone: the lane stays a candidate and the exact test decides)
                        }
                    }
                    bool cand = false;
#pragma unroll
                    for (int i = 0; i < NR; ++i)
#pragma unroll
                        for (int q = 0; q < VEC; ++q) cand = cand || (mx[i][q] <= thr2);
                    if (__any(cand)) {  // the exact test of noise.jl:31-37, as the interpreter does it
#pragma unroll
                        for (int i = 0; i < NR; ++i)
#pragma unroll
                            for (int q = 0; q < VEC; ++q) {
                                bool same = true;
                                for (int k = 0; k < d; ++k) {
                                    if ((am >> k) & 1) {
                                        const T a = ar[(rr + i) * DMAX + k], b = xbs[(col0 + (int64_t)lane * VEC + q) * DMAX + k];
                                        const T m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
                                        same = same && ((a == b) || (fabs(a - b) <= Tr<T>::isapprox_rtol * m));
                                    }
                                }
                                val[i][q] = same ? sig2 : T(0);
                            }
                    }
                }
            } else {
                T r[NR][VEC];
#pragma unroll
                for (int i = 0; i < NR; ++i)
#pragma unroll
                    for (int q = 0; q < VEC; ++q) r[i][q] = T(0);
#pragma unroll
                for (int k = 0; k < DMAX; ++k) {
                    const T wk = s_leaf[o].w[k];  // LDS broadcast
#pragma unroll
                    for (int i = 0; i < NR; ++i)
#pragma unroll
                        for (int q = 0; q < VEC; ++q) r[i][q] = Tr<T>::fma_(dsq[i][k][q], wk, r[i][q]);
                }
                const T p0inv = s_leaf[o].p0, p1 = s_leaf[o].p1;
                (void)p1;
                switch (op) {  // the leaf files' formulas (include/gpmi.h), r = the weighted squared distance
                    case GPMI_K_SE_ISO:
                    case GPMI_K_SE_ARD:
#pragma unroll
                        for (int i = 0; i < NR; ++i)
#pragma unroll
                            for (int q = 0; q < VEC; ++q) val[i][q] = sig2 * Tr<T>::exp_((T(-0.5) * r[i][q]) * p0inv);
                        break;
                    case GPMI_K_MAT12_ISO:
                    case GPMI_K_MAT12_ARD:
#pragma unroll
                        for (int i = 0; i < NR; ++i)
#pragma unroll
                            for (int q = 0; q < VEC; ++q) val[i][q] = sig2 * Tr<T>::exp_(-(sqrt_nonneg(r[i][q]) * p0inv));
                        break;
                    case GPMI_K_MAT32_ISO:
                    case GPMI_K_MAT32_ARD:
#pragma unroll
                        for (int i = 0; i < NR; ++i)
#pragma unroll
                            for (int q = 0; q < VEC; ++q) {
                                const T sv = T(1.7320508075688772935) * sqrt_nonneg(r[i][q]) * p0inv;
                                val[i][q] = sig2 * (T(1) + sv) * Tr<T>::exp_(-sv);
                            }
                        break;
                    case GPMI_K_MAT52_ISO:
                    case GPMI_K_MAT52_ARD:
#pragma unroll
                        for (int i = 0; i < NR; ++i)
#pragma unroll
                            for (int q = 0; q < VEC; ++q) {
                                const T sv = T(2.2360679774997896964) * sqrt_nonneg(r[i][q]) * p0inv;
                                val[i][q] = sig2 * (T(1) + sv + sv * sv * T(1.0 / 3.0)) * Tr<T>::exp_(-sv);
                            }
                        break;
                    default:  // GPMI_K_RQ_*
                        if constexpr (FEAT & 1) {
#pragma unroll
                            for (int i = 0; i < NR; ++i)
#pragma unroll
                                for (int q = 0; q < VEC; ++q) val[i][q] = sig2 * Tr<T>::pow_(T(1) + r[i][q] * p0inv, -p1);
                        } else {
#pragma unroll
                            for (int i = 0; i < NR; ++i)
#pragma unroll
                                for (int q = 0; q < VEC; ++q) val[i][q] = sig2;
                        }
                        break;
                }
            }
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int q = 0; q < VEC; ++q) {  // push
                    s2[i][q] = s1[i][q];
                    s1[i][q] = s0[i][q];
                    s0[i][q] = val[i][q];
                }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            VT out;
#pragma unroll
            for (int q = 0; q < VEC; ++q) out[q] = s0[i][q];
            *reinterpret_cast<VT*>(crow + (int64_t)(rr + i) * ldc) = out;
        }
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
    // interior, off-diagonal tiles of single-leaf and shallow multi-leaf programs belong to cov_leaf_kernel / cov_multi_kernel
    // (launched alongside): leave before staging anything — this kernel keeps the diagonal, edge and padded tiles
    if constexpr (DMAX > 0) {
        if (fast_tile(prog, prog->n_ops, flags, row0, col0, TR, TC, na, nb, nrows, ncols, row_off)) return;
    }

    const int tid = threadIdx.x;
    // d beyond what two staged blocks of LDS hold (COV_GLOBAL_X, DMAX == 0 only): the operands are read from global memory where
    // they are used (the column block through L1 / L2, the row values wave-uniform) — any input dimension, as the reference's
    // distance loops (src/kernels/distance.jl:41-106); slower per entry, and rare
    const bool gx = DMAX == 0 && (flags & COV_GLOBAL_X);
    if (!gx) {
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
    }

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


    for (int rr = 0; rr < TR / 4; ++rr) {
        const int row = wave * (TR / 4) + rr;
        const int64_t grow = row0 + row;
        if (grow >= nrows) break;
        const T* sar = sa + row * d;
        // global-operand form: clamped row / column pointers (padding rows repeat the last point; they are masked when stored)
        int64_t grc = grow < na ? grow : na - 1;
        grc = grc < 0 ? 0 : grc;
        const T* gar = xa + grc * d;
        const T* gbc[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            int64_t gc = col0 + (int64_t)lane * VEC + q;
            gc = gc < nb ? gc : nb - 1;
            gbc[q] = xb + gc * d;
        }

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
                const double* w = prog->wtab() + prog->leaf[o].woff;
                bool same[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) same[q] = true;
                for (int k = 0; k < d; ++k) {
                    if (w[k] != 0.0) {
                        T a = gx ? gar[k] : sar[k];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) {
                            T b = gx ? gbc[q][k] : sbT[k * TC + lane * VEC + q];
                            T m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
                            bool ok = (a == b) || (fabs(a - b) <= Tr<T>::isapprox_rtol * m);
                            same[q] = same[q] && ok;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < VEC; ++q) val[q] = same[q] ? s2 : T(0);
            } else {
                const double* w = prog->wtab() + prog->leaf[o].woff;
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
                        T a = gx ? gar[k] : sar[k];
                        VT bv;
                        if (gx) {
#pragma unroll
                            for (int q = 0; q < VEC; ++q) bv[q] = gbc[q][k];
                        } else {
                            bv = *reinterpret_cast<const VT*>(&sbT[k * TC + lane * VEC]);
                        }
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
    if (DMAX == 0 && lds > 128 * 1024) {  // the two staged blocks do not fit the LDS: operands from global memory (any d)
        lds = 0;
        flags |= COV_GLOBAL_X;
    }
    auto kern = cov_kernel<T, DMAX>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const DevProgram* hp = ctx->h_prog;
    bool single_leaf = false, multi = false;
    T *xas = nullptr, *xbs = nullptr;
    unsigned long long* amax = nullptr;
    if constexpr (DMAX > 0) {
        single_leaf = hp->n_ops == 1 && hp->leaf[0].op != GPMI_K_NOISE && hp->leaf[0].op != GPMI_K_CONST;
        multi = hp->n_ops > 1 && hp->fast_class >= 0;
        // both specialised kernels work on zero-padded (single leaf: also pre-scaled) copies of the two input blocks, 16 bytes of header
        // (max |x|) in front: without the scratch the interpreter takes every tile
        if ((single_leaf || multi) && grow(ctx, &ctx->cov_scaled, &ctx->cov_scaled_cap, 16 + (na + nb) * DMAX * (int64_t)sizeof(T)) != GPMI_OK) {
            (void)hipGetLastError();
            single_leaf = multi = false;
            flags |= COV_NO_FAST;
        }
        if (single_leaf || multi) {
            amax = (unsigned long long*)ctx->cov_scaled;
            xas = (T*)((char*)ctx->cov_scaled + 16);
            xbs = xas + na * DMAX;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, xa, na, xb, nb, d, C, ldc, nrows_total, ncols_total,
                       ctx->d_prog, flags, nugget, nugget_vec, row_off);
    if constexpr (DMAX > 0) {
        if (single_leaf) {
            // pre-scaled, zero-padded copies of the two input blocks (na d + nb d elements), then the family's kernel
            const DevLeaf& lf = hp->leaf[0];
            int fam = FAM_SE;
            double mult = lf.p0;  // SE iso: 1/l2; RQ: 1/(2 a l2) | 0.5/a; ARD SE: 1
            switch (lf.op) {
                case GPMI_K_MAT12_ISO: case GPMI_K_MAT12_ARD: fam = FAM_MAT12; mult = lf.p0 * lf.p0; break;
                case GPMI_K_MAT32_ISO: case GPMI_K_MAT32_ARD: fam = FAM_MAT32; mult = 3.0 * lf.p0 * lf.p0; break;
                case GPMI_K_MAT52_ISO: case GPMI_K_MAT52_ARD: fam = FAM_MAT52; mult = 5.0 * lf.p0 * lf.p0; break;
                case GPMI_K_RQ_ISO: case GPMI_K_RQ_ARD: fam = FAM_RQ; break;
                default: break;
            }
            ScaleW<DMAX> sw;
            for (int k = 0; k < DMAX; ++k) sw.sw[k] = k < d ? sqrt(hp->wtab()[lf.woff + k] * mult) : 0.0;
            {
                if (na > 0)
                    hipLaunchKernelGGL((scale_inputs_kernel<T, DMAX>), dim3((unsigned)((na * DMAX + 255) / 256)), dim3(256), 0, ctx->stream, xa, na, d, sw, xas,
                                       (unsigned long long*)nullptr, nb > 0 ? xb : (const T*)nullptr /* both blocks centred on the first point of xb */);
                hipLaunchKernelGGL((scale_inputs_kernel<T, DMAX>), dim3((unsigned)((nb * DMAX + 255) / 256)), dim3(256), 0, ctx->stream, xb, nb, d, sw, xbs,
                                   (unsigned long long*)nullptr, xb);
                auto go = [&](auto lk) {
                    hipLaunchKernelGGL(lk, grid, dim3(256), 0, ctx->stream, (const T*)xas, na, (const T*)xbs, nb, C, ldc, nrows_total, ncols_total,
                                       ctx->d_prog, flags, row_off, (T)lf.s2, (T)lf.p1);
                };
                switch (fam) {
                    case FAM_SE: go(cov_leaf_kernel<T, DMAX, FAM_SE>); break;
                    case FAM_MAT12: go(cov_leaf_kernel<T, DMAX, FAM_MAT12>); break;
                    case FAM_MAT32: go(cov_leaf_kernel<T, DMAX, FAM_MAT32>); break;
                    case FAM_MAT52: go(cov_leaf_kernel<T, DMAX, FAM_MAT52>); break;
                    default: go(cov_leaf_kernel<T, DMAX, FAM_RQ>); break;
                }
            }
        } else if (multi) {
            ScaleW<DMAX> one;  // unit weights: the copies are only zero-padded to DMAX
            for (int k = 0; k < DMAX; ++k) one.sw[k] = 1.0;
            (void)hipMemsetAsync(amax, 0, 16, ctx->stream);
            if (na > 0)
                hipLaunchKernelGGL((scale_inputs_kernel<T, DMAX>), dim3((unsigned)((na * DMAX + 255) / 256)), dim3(256), 0, ctx->stream, xa, na, d, one, xas, amax, (const T*)nullptr);
            hipLaunchKernelGGL((scale_inputs_kernel<T, DMAX>), dim3((unsigned)((nb * DMAX + 255) / 256)), dim3(256), 0, ctx->stream, xb, nb, d, one, xbs, amax, (const T*)nullptr);
            auto go = [&](auto mk) {
                hipLaunchKernelGGL(mk, grid, dim3(256), 0, ctx->stream, (const T*)xas, na, (const T*)xbs, nb, C, ldc, nrows_total, ncols_total, ctx->d_prog,
                                   flags, row_off, (const unsigned long long*)amax);
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
    // (rows [row_off, row_off + nrows) of the lower region: row i keeps its first min(i + 1, ncols) columns)
    double entries = (double)nrows_total * (double)ncols_total;
    if (flags & COV_LOWER) {
        const double r0 = (double)row_off, nr = (double)nrows_total, nc = (double)ncols_total;
        const double tri = std::max(0.0, std::min(nr, nc - r0));  // rows still under the diagonal
        entries = tri * (r0 + 0.5 * (tri + 1.0)) + (nr - tri) * nc;
    }
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
