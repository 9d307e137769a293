// potf2.h — the 64 x 64 diagonal-block Cholesky + explicit inverse executed by ONE wavefront (the POTF2 of the blocked dpotrf behind
// make_posdef!, src/GP.jl:110), shared by diag64_kernel (panel.hip) and the persistent chain kernel (chain.hip).  Device code only.
#pragma once
#include "common.h"
#include "mfma.h"

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
// The reciprocals 1 / L_jj are kept in `invdiag`.
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
__device__ __forceinline__ void store16(const typename Mfma<T>::Acc& acc, T* C, int ldc, T scale, bool transpose, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = Mfma<T>::row_of(lane, r), col = Mfma<T>::col_of(lane, r);
        const T v = scale * acc_get<T>(acc, r);
        if (transpose)
            C[col * ldc + row] = v;
        else
            C[row * ldc + col] = v;
    }
}

// diag64: Cholesky of one 64 x 64 diagonal block AND its explicit inverse, ONE wavefront.
//   1. potf2 with the rows in registers, blocked by 16 columns (rank-16 updates on the matrix cores);
//   2. the four 16 x 16 diagonal blocks of L are inverted by per-lane substitution (16 lanes per block);
//   3. the off-diagonal blocks of L^-1 follow from  X_ib = -Dinv_i * sum_{t=b}^{i-1} L_it X_tb  on the matrix
//      cores (16 tiny products, LDS-resident operands);
// so that every later solve against this block (panel TRSM, whiten!, back-substitution) is a GEMM.
// LDS pool of the panel kernels, in elements: diag64 needs S + XT + DI + WT + sinv, rows64 two 64 x 65 buffers (a subset)
constexpr int PANEL_POOL = 2 * 64 * 65 + 64 * 16 + 16 * 17 + 64;

// phase clock of the trace build (tools/chain_trace.py; never in libgpmi.so): the 100 MHz clock at six points of the LAST potf2 that ran
#ifdef GPMI_CHAIN_TRACE
__device__ unsigned long long g_potf2_marks[8];
#define POTF2_MARK(slot)                                                         \
    do {                                                                         \
        if ((threadIdx.x & 63) == 0) g_potf2_marks[slot] = wall_clock64();       \
    } while (0)
#else
#define POTF2_MARK(slot) ((void)0)
#endif

// barrier of a phase that ONE wavefront executes (diag64): LDS operations of a wave complete in issue order, so a fence that
// keeps the compiler from moving them (and waits for them) is all a single wave needs — and, unlike __syncthreads(), it does
// not involve the workgroup's other waves
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// FROM_LDS (chain.hip): the block is already in S (pool, row-major, leading dimension 65) and the results STAY in LDS — S = L (strict upper
// part zero), XT[n][k] = Linv[k][n] — for the caller's cooperative stores; only invdiag goes to global memory here.  Returns the failing
// pivot (1-based within the block) or 0, wave-uniform; on failure nothing is written but *info.
template <typename T, bool FROM_LDS = false>
__device__ __forceinline__ int diag64_body(T* __restrict__ A, int64_t ld, T* __restrict__ Linv, T* __restrict__ invdiag,
                                           int* __restrict__ info, int64_t pivot_base, T* __restrict__ pool) {
    constexpr int SLD = 65;
    T* const S = pool;                  // [64 * SLD]  L, row-major
    T* const XT = S + 64 * SLD;         // [64 * SLD]  XT[n][k] = Linv[k][n]
    T* const DI = XT + 64 * SLD;        // [64 * 16]   DI[16 b + i][c] = (L_bb^-1)[i][c]
    T* const WT = DI + 64 * 16;         // [16 * 17]
    T* const sinv = WT + 16 * 17;       // [64]
    const int i = threadIdx.x & 63;
    if constexpr (!FROM_LDS) {
        for (int r = 0; r < 64; ++r) S[r * SLD + i] = A[(int64_t)r * ld + i];  // coalesced rows
        wave_sync();
    }
    POTF2_MARK(0);
    T a[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) a[c] = S[i * SLD + c];

    // Blocked by 16 columns: inside a block the rank-1 updates stay in registers (<= 15 per step); the rank-16
    // update of everything to the right goes through the matrix cores (panel -> LDS -> P P' tiles -> LDS -> rows).
    int fail = 0;
    using AccP = typename Mfma<T>::Acc;
    T* PL = XT;  // [64][17] panel image (XT is not needed before the inverse phase)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * b + jj;
            const T d = bcast_lane<T>(a[j], j);
            // a non-positive (or NaN) pivot is recorded once; the remaining steps run on garbage and are discarded
            // (no early exit: every index into a[] must stay a compile-time constant to keep the rows in registers)
            if (fail == 0 && !(d > T(0))) fail = j + 1;
            T r = rsqrt_seed<T>(d);
            r = r * (T(1.5) - T(0.5) * d * r * r);
            r = r * (T(1.5) - T(0.5) * d * r * r);
            T sq = d * r;
            sq = sq + (T(0.5) * r) * (d - sq * sq);       // sqrt(d)
            const T inv = r + r * (T(1) - sq * r);        // 1 / sqrt(d)
            const T lij = (i == j) ? sq : a[j] * inv;
            a[j] = lij;
            // 1 / L_jj is the same in every lane: lane 0 parks it in LDS.  (Kept per lane as `if (i == j) myinv = inv`, the compiler held all
            // 64 reciprocals live to the end of the loop and selected there — spills, and a chain of scratch reloads worth ~3 us of the 22.)
            if (i == 0) sinv[j] = inv;
#pragma unroll
            for (int c = j + 1; c < 16 * b + 16; ++c) a[c] -= lij * bcast_lane<T>(lij, c);
        }
        if (b < 3) {
            wave_sync();
#pragma unroll
            for (int q = 0; q < 16; ++q) PL[i * 17 + q] = a[16 * b + q];
            wave_sync();
            for (int ri = b + 1; ri < 4; ++ri)
                for (int ci = b + 1; ci <= ri; ++ci) {
                    AccP u;
                    acc_zero<T>(u);
                    mma16_nt<T>(u, PL + 16 * ri * 17, 17, PL + 16 * ci * 17, 17, 16, i);
                    store16<T>(u, S + (16 * ri) * SLD + 16 * ci, SLD, T(1), false, i);
                }
            wave_sync();
#pragma unroll
            for (int ci = b + 1; ci < 4; ++ci) {
                if (ci <= (i >> 4)) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) a[16 * ci + q] -= S[i * SLD + 16 * ci + q];
                }
            }
        }
    }
    POTF2_MARK(1);  // the 64 columns done
    if (fail) {
        if (i == 0) {
            if constexpr (FROM_LDS) {
                // chain.hip: tasks already in flight when a pivot fails run on with garbage, and a LATER diagonal tile may fail on it: the
                // first failure (diagonal tiles finish in order: the smallest pivot) must stay — dpotrf's info
                atomicCAS(info, 0, (int)(pivot_base + fail));
            } else {
                *info = (int)(pivot_base + fail);
            }
        }
        return fail;
    }
    wave_sync();
    invdiag[i] = sinv[i];
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        S[i * SLD + c] = (c <= i) ? a[c] : T(0);
        XT[i * SLD + c] = T(0);
    }
    wave_sync();
    if constexpr (!FROM_LDS)
        for (int r = 0; r < 64; ++r) A[(int64_t)r * ld + i] = S[r * SLD + i];  // lower triangle = L, strict upper = 0

    POTF2_MARK(2);  // L (and the zeroed XT) in LDS
    // ---- 16 x 16 diagonal inverses: lane (b, c) computes column c of (L_bb)^-1 ----------------------------
    {
        const int b = i >> 4, c = i & 15;
        T x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            T sacc = (q == c) ? T(1) : T(0);
#pragma unroll
            for (int t = 0; t < q; ++t) sacc -= S[(16 * b + q) * SLD + 16 * b + t] * x[t];
            x[q] = sacc * sinv[16 * b + q];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            DI[(16 * b + q) * 16 + c] = x[q];
            XT[(16 * b + c) * SLD + 16 * b + q] = x[q];
        }
    }
    wave_sync();
    POTF2_MARK(3);  // the four 16 x 16 inverses
    // ---- off-diagonal blocks, by distance from the diagonal ---------------------------------------------
    using Acc = typename Mfma<T>::Acc;
    for (int dist = 1; dist < 4; ++dist) {
        for (int b = 0; b + dist < 4; ++b) {
            const int ib = b + dist;
            Acc w;
            acc_zero<T>(w);
            for (int t = b; t < ib; ++t)  // W = sum_t L_it X_tb ;  B operand rows n: XT[16 b + n][16 t + k]
                mma16_nt<T>(w, S + (16 * ib) * SLD + 16 * t, SLD, XT + (16 * b) * SLD + 16 * t, SLD, 16, i);
            store16<T>(w, WT, 17, T(1), true, i);  // WT[n][k] = W[k][n]
            wave_sync();
            Acc xacc;
            acc_zero<T>(xacc);
            mma16_nt<T>(xacc, DI + (16 * ib) * 16, 16, WT, 17, 16, i);  // Dinv_i * W
            store16<T>(xacc, XT + (16 * b) * SLD + 16 * ib, SLD, T(-1), true, i);  // XT[16b + n][16 ib + row] = -X[row][n]
            wave_sync();
        }
    }
    POTF2_MARK(4);  // the six off-diagonal blocks
    if constexpr (!FROM_LDS)
        for (int k = 0; k < 64; ++k) Linv[k * 64 + i] = XT[i * SLD + k];  // Linv[k][n] = XT[n][k]
    return 0;
}

}  // namespace
}  // namespace gpmi
