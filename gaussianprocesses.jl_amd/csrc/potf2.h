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


// =============================================================================================================================
// potf2_wg — the same 64 x 64 diagonal block (Cholesky factor + explicit inverse) by a whole 256-thread WORKGROUP (chain.hip's diagonal
// task; round 6).  diag64_body above walks 64 dependent column steps on one wavefront with the other three idle (21.4 us measured
// inside the chain launch: 64 columns 9.9 — the compiler turns the in-register rank-1 updates into a serial chain of up to 15 dependent
// FMAs in front of every pivot —, 128 LDS stores 3.6, the four 16 x 16 substitutions 1.4, six off-diagonal inverse blocks 3.5, ~3 of
// callee-saved spills around 128 row registers).  Here the block lives in LDS and is walked in four 16-column PANELS:
//   P_b (wave 0): the panel's 16 columns in 16 registers per lane (lane = row), right-looking inside the panel.  Per column ONE chain
//        pivot -> v_rsq + two Newton steps (1 / sqrt(d) only: sqrt(d) = d r falls out of the scaling) -> r^2 -> the next pivot column's
//        FMA; everything else of the step — the 15 broadcasts of the UNSCALED column (they do not need r), the other columns' FMAs —
//        is independent work for the latency bubbles of that chain.
//        For b >= 1 lanes 0..15 (rows above the panel: idle) carry the rows of a 16 x 16 IDENTITY through the same column operations:
//        the operations amount to a right-multiplication by L_bb^-T, so those lanes end up holding the rows of L_bb^-T — the 16 x 16
//        diagonal inverse at no instruction of its own, and off the critical path.  (Block 0 has no idle lanes: wave 1 does its
//        substitution while wave 0 is in panel 1.)
//   U_b (waves 1..3, one tile each): the rank-16 update of the NEXT panel's tiles on the matrix cores; wave 0 goes on as soon as those
//        are done, the tiles further right are updated by waves 1..3 UNDER panel b+1.
//   inverse: X_ib = -D_i sum_t L_it X_tb block by block on waves 1..3 as their operands become final, also under the panels; after the
//        last panel only X_3b = -D_3 W_3b (three waves side by side) is left.
// Eight workgroup barriers in all.  Input: the symmetric block in S = pool (row-major, leading dimension 65).  Output, in LDS for the
// caller's cooperative stores: S = L (strict upper part zero), XT[n][k] = Linv[k][n]; invdiag to global memory.  Returns the failing pivot
// (1-based within the block) or 0, the same value in every thread; on failure *info is set (first failure wins) and S / XT are garbage.
// =============================================================================================================================
constexpr int POTF2_WG_POOL = 2 * 64 * 65 + 64 * 16 + 3 * 16 * 17;  // S, XT, DI, one WT per side wave

template <typename T>
__device__ __forceinline__ T tfma(T a, T b, T c);
template <>
__device__ __forceinline__ double tfma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <>
__device__ __forceinline__ float tfma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <typename T>
__device__ __forceinline__ T rcp_seed(T x);
template <>
__device__ __forceinline__ double rcp_seed<double>(double x) { return __builtin_amdgcn_rcp(x); }
template <>
__device__ __forceinline__ float rcp_seed<float>(float x) { return __builtin_amdgcn_rcpf(x); }

// 1 / sqrt(d): hardware seed (v_rsq_f64: ~2^-23 relative) + two Newton steps r += r (1/2 - (d/2) r^2) in fused form: ~1 ulp, six dependent
// operations behind the seed.  d <= 0, NaN, inf give NaN or inf — the caller's pivot test reads it off the stored diagonal.
template <typename T>
__device__ __forceinline__ T rsqrt_newton2(T d) {
    T r = rsqrt_seed<T>(d);
    const T hd = T(0.5) * d;
    T e = tfma<T>(-(hd * r), r, T(0.5));
    r = tfma<T>(r, e, r);
    e = tfma<T>(-(hd * r), r, T(0.5));
    r = tfma<T>(r, e, r);
    return r;
}
template <typename T>
__device__ __forceinline__ T rcp_newton2(T x) {
    T r = rcp_seed<T>(x);
    r = tfma<T>(r, tfma<T>(-x, r, T(1)), r);
    r = tfma<T>(r, tfma<T>(-x, r, T(1)), r);
    return r;
}

// P_b: wave 0.  Compile-time panel index: every lane index of a broadcast is an immediate.
// The 16 column steps are SOFTWARE-PIPELINED by hand (hipcc's schedule of the plain loop puts the step's 2 x 15 broadcasts and 14 FMAs
// between one pivot's chain and the next — on the critical path in issue order): step jj issues its chain
//     v_rsq(d), t = (d/2) r, e = 1/2 - t r, r += r e, t, e, r, r^2, next pivot column's FMA, broadcast of the next pivot
// as ten instructions with a slice of INDEPENDENT work behind each of the first seven — the FMAs that step jj-1 owes columns jj+1..15
// and this step's broadcasts of the unscaled column jj — fenced by sched_barriers so that the order written is the order issued.  A step
// then costs max(chain latency, issue slots) instead of their sum.
template <typename T, int B, bool PIPE = true>
__device__ __forceinline__ void potf2_panel(T* __restrict__ S, T* __restrict__ XT, T* __restrict__ DI, int lane) {
    constexpr int SLD = 65, base = 16 * B;
    T a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = S[lane * SLD + base + c];
    // (B > 0: the rows above the panel arrive from S as the identity in lanes 0..15 and zeros in the others — potf2_wg plants them in phase 0)
    if constexpr (PIPE) {
#define POTF2_SB() __builtin_amdgcn_sched_barrier(0)
        T sp[16], sc[16];  // broadcasts (wave-uniform: scalar registers) of the previous / the current step's unscaled column
        T lrp = T(0);      // a_ij r^2 of the previous step
        T d = bcast_lane<T>(a[0], base);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int nF = jj >= 1 ? 15 - jj : 0, nR = 15 - jj, n = nF + nR;
            // item k of the step's independent work: F(jj+1), R(jj+1), F(jj+2), R(jj+2), ...   (jj = 0: broadcasts only)
            auto items = [&](int g) __attribute__((always_inline)) {
#pragma unroll
                for (int k = g * n / 7; k < (g + 1) * n / 7; ++k) {
                    const bool isF = jj >= 1 && (k & 1) == 0;
                    const int c = jj + 1 + (jj >= 1 ? (k >> 1) : k);
                    if (isF)
                        a[c] = tfma<T>(-lrp, sp[c], a[c]);
                    else
                        sc[c] = bcast_lane<T>(a[jj], base + c);
                }
            };
            POTF2_SB();
            T r = rsqrt_seed<T>(d);
            const T hd = T(0.5) * d;
            POTF2_SB(); items(0); POTF2_SB();
            T t = hd * r;
            POTF2_SB(); items(1); POTF2_SB();
            T e = tfma<T>(-t, r, T(0.5));
            POTF2_SB(); items(2); POTF2_SB();
            r = tfma<T>(r, e, r);
            POTF2_SB(); items(3); POTF2_SB();
            t = hd * r;
            POTF2_SB(); items(4); POTF2_SB();
            e = tfma<T>(-t, r, T(0.5));
            POTF2_SB(); items(5); POTF2_SB();
            r = tfma<T>(r, e, r);
            POTF2_SB(); items(6); POTF2_SB();
            const T rr = r * r;
            if (jj < 15) {
                // (every item of the step has been issued by now — sc[jj + 1] may come with the last slice when the step has few items)
                const T p = a[jj] * sc[jj + 1];
                a[jj + 1] = tfma<T>(-p, rr, a[jj + 1]);  // the next pivot's column: two operations behind r
                d = bcast_lane<T>(a[jj + 1], base + jj + 1);
            }
            lrp = a[jj] * rr;
            a[jj] = a[jj] * r;  // lane j: d r = sqrt(d)
#pragma unroll
            for (int c = jj + 2; c < 16; ++c) sp[c] = sc[c];
        }
        POTF2_SB();
#undef POTF2_SB
    } else {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const T d = bcast_lane<T>(a[jj], base + jj);
            // the column's entries in the panel's own rows, UNSCALED: l_cj = s[c] r — none of these broadcasts waits for r
            T s[16];
#pragma unroll
            for (int c = jj + 1; c < 16; ++c) s[c] = bcast_lane<T>(a[jj], base + c);
            T p = T(0);
            if (jj < 15) p = a[jj] * s[jj + 1];
            const T r = rsqrt_newton2<T>(d);
            const T rr = r * r;
            if (jj < 15) {
                a[jj + 1] = tfma<T>(-p, rr, a[jj + 1]);  // the next pivot's column first: two operations behind r
                asm volatile("" : "+v"(a[jj + 1]));      // (pinned: hipcc otherwise sinks every update to just in front of its column's pivot — a serial FMA chain)
            }
            const T l = a[jj] * r;   // lane j: d r = sqrt(d)
            const T lr = a[jj] * rr;
#pragma unroll
            for (int c = jj + 2; c < 16; ++c) {
                a[c] = tfma<T>(-lr, s[c], a[c]);
                asm volatile("" : "+v"(a[c]));
            }
            a[jj] = l;
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) S[lane * SLD + base + c] = (base + c <= lane) ? a[c] : T(0);
    if constexpr (B > 0) {
        if (lane < 16) {  // e[c] = (L_bb^-T)[lane][c]:  XT's diagonal block row `lane`, DI's (= L_bb^-1, row-major 16-wide) column `lane`
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                XT[(base + lane) * SLD + base + c] = a[c];
                DI[(base + c) * 16 + lane] = a[c];
            }
        }
    }
}

// S(ri, ci) -= L(ri, b) L(ci, b)' for one 16 x 16 tile, one wave
template <typename T>
__device__ __forceinline__ void potf2_update_tile(T* __restrict__ S, int ri, int ci, int b, int lane) {
    constexpr int SLD = 65;
    typename Mfma<T>::Acc u;
    acc_zero<T>(u);
    mma16_nt<T>(u, S + (16 * ri) * SLD + 16 * b, SLD, S + (16 * ci) * SLD + 16 * b, SLD, 16, lane);
    T* C = S + (16 * ri) * SLD + 16 * ci;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = Mfma<T>::row_of(lane, r), col = Mfma<T>::col_of(lane, r);
        C[row * SLD + col] -= acc_get<T>(u, r);
    }
}
// w += sum_{t = t0}^{t1 - 1} L(ib, t) X(t, bb)      (B operand rows n: XT[16 bb + n][16 t + k] = X_tb[k][n])
template <typename T>
__device__ __forceinline__ void potf2_w_accum(typename Mfma<T>::Acc& w, const T* __restrict__ S, const T* __restrict__ XT, int ib, int bb, int t0, int t1,
                                              int lane) {
    constexpr int SLD = 65;
    for (int t = t0; t < t1; ++t) mma16_nt<T>(w, S + (16 * ib) * SLD + 16 * t, SLD, XT + (16 * bb) * SLD + 16 * t, SLD, 16, lane);
}
// X(ib, bb) = -D_ib w  ->  XT[16 bb + n][16 ib + row];  WTw: this wave's 16 x 17 staging buffer
template <typename T>
__device__ __forceinline__ void potf2_x_from_w(const typename Mfma<T>::Acc& w, T* __restrict__ XT, const T* __restrict__ DI, T* __restrict__ WTw, int ib,
                                               int bb, int lane) {
    constexpr int SLD = 65;
    store16<T>(w, WTw, 17, T(1), true, lane);  // WT[n][k] = W[k][n]
    wave_sync();
    typename Mfma<T>::Acc x;
    acc_zero<T>(x);
    mma16_nt<T>(x, DI + (16 * ib) * 16, 16, WTw, 17, 16, lane);
    store16<T>(x, XT + (16 * bb) * SLD + 16 * ib, SLD, T(-1), true, lane);
    wave_sync();
}

template <typename T>
__device__ __forceinline__ int potf2_wg(T* __restrict__ invdiag, int* __restrict__ info, int64_t pivot_base, T* __restrict__ pool) {
    constexpr int SLD = 65;
    T* const S = pool;                  // [64 * SLD]
    T* const XT = S + 64 * SLD;         // [64 * SLD]
    T* const DI = XT + 64 * SLD;        // [64 * 16]   DI[16 b + i][c] = (L_bb^-1)[i][c]
    T* const WT = DI + 64 * 16;         // [3][16 * 17]
    __shared__ int s_fail;
    using Acc = typename Mfma<T>::Acc;
    // (opaque: everything below that is computed from the lane number — dozens of LDS addresses — is loop-invariant in the caller's task
    //  loop; hipcc hoists it all out, runs out of registers and reloads the addresses from scratch in the middle of the panel phases)
    int tid_o = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_o));
    const int lane = tid_o & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid_o >> 6);
    T* const WTw = WT + (wv > 0 ? wv - 1 : 0) * (16 * 17);
    Acc w;  // waves 2 / 3 / 1: a W block carried from the phase that can form it to the phase that has its D
    acc_zero<T>(w);

#ifdef GPMI_CHAIN_TRACE
#define POTF2_WG_MARK(slot) do { if (tid_o == 0) g_potf2_marks[slot] = wall_clock64(); } while (0)
#else
#define POTF2_WG_MARK(slot) ((void)0)
#endif
    POTF2_WG_MARK(0);
    // ---- phase 0: panel 0 | XT zeroed --------------------------------------------------------------------------------------------
    if (wv == 0) {
        potf2_panel<T, 0>(S, XT, DI, lane);
    } else {
        for (int e = tid_o - 64; e < 64 * SLD; e += 192) XT[e] = T(0);
        // the rows ABOVE panels 1..3 (the symmetric input's upper part: read by nobody): a 16 x 16 identity in rows 0..15 of every panel's
        // columns, zeros below it down to the panel's own rows — what lanes 0 .. 16 b - 1 of wave 0 then simply load in P_b
        for (int e = tid_o - 64; e < 48 * 48; e += 192) {
            const int q = e / 48, col = 16 + e % 48;  // row q, column col >= 16
            if (q < (col & ~15)) S[q * SLD + col] = (q == (col & 15)) ? T(1) : T(0);
        }
    }
    __syncthreads();
    if (wv >= 1) potf2_update_tile<T>(S, wv, 1, 0, lane);  // tiles (1,1) (2,1) (3,1)
    __syncthreads();
    // ---- phase 1: panel 1 | D_0 by substitution; panel 0's updates of (2,2) (3,2) (3,3) ------------------------------------------
    if (wv == 0) {
        potf2_panel<T, 1>(S, XT, DI, lane);
    } else if (wv == 1) {
        // column c of L_00^-1 per lane (lanes 16..63 repeat lanes 0..15), right-looking: x_q final -> every later entry gets its term at once
        int c = lane & 15;
        asm volatile("" : "+v"(c));  // (opaque: hipcc otherwise builds the unit vectors below once per kernel, outside the task loop, and spills them)
        const T ri = rcp_newton2<T>(S[c * SLD + c]);
        T x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] = (t == c) ? T(1) : T(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            x[q] *= bcast_lane<T>(ri, q);
#pragma unroll
            for (int t = q + 1; t < 16; ++t) x[t] = tfma<T>(-S[t * SLD + q], x[q], x[t]);  // (one address for the whole wave: an LDS broadcast)
        }
        if (lane < 16) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                DI[q * 16 + c] = x[q];
                XT[c * SLD + q] = x[q];
            }
        }
    } else if (wv == 2) {
        potf2_update_tile<T>(S, 2, 2, 0, lane);
    } else {
        potf2_update_tile<T>(S, 3, 2, 0, lane);
        potf2_update_tile<T>(S, 3, 3, 0, lane);
    }
    __syncthreads();
    if (wv == 1) potf2_update_tile<T>(S, 2, 2, 1, lane);
    if (wv == 2) potf2_update_tile<T>(S, 3, 2, 1, lane);
    __syncthreads();
    // ---- phase 2: panel 2 | panel 1's update of (3,3); X_10, then W_20 and W_21 (their D_2 comes with this panel) ----------------
    if (wv == 0) {
        potf2_panel<T, 2>(S, XT, DI, lane);
    } else if (wv == 1) {
        potf2_update_tile<T>(S, 3, 3, 1, lane);
    } else if (wv == 2) {
        potf2_w_accum<T>(w, S, XT, 1, 0, 0, 1, lane);
        potf2_x_from_w<T>(w, XT, DI, WTw, 1, 0, lane);   // X_10
        acc_zero<T>(w);
        potf2_w_accum<T>(w, S, XT, 2, 0, 0, 2, lane);    // W_20 = L_20 X_00 + L_21 X_10
    } else {
        potf2_w_accum<T>(w, S, XT, 2, 1, 1, 2, lane);    // W_21 = L_21 X_11
    }
    __syncthreads();
    if (wv == 1) potf2_update_tile<T>(S, 3, 3, 2, lane);
    __syncthreads();
    // ---- phase 3: panel 3 | X_20, X_21; W_30, W_31, W_32 --------------------------------------------------------------------------
    if (wv == 0) {
        potf2_panel<T, 3>(S, XT, DI, lane);
    } else if (wv == 1) {
        potf2_w_accum<T>(w, S, XT, 3, 2, 2, 3, lane);    // W_32 = L_32 X_22
    } else if (wv == 2) {
        potf2_x_from_w<T>(w, XT, DI, WTw, 2, 0, lane);   // X_20
        acc_zero<T>(w);
        potf2_w_accum<T>(w, S, XT, 3, 0, 0, 3, lane);    // W_30
    } else {
        potf2_x_from_w<T>(w, XT, DI, WTw, 2, 1, lane);   // X_21
        acc_zero<T>(w);
        potf2_w_accum<T>(w, S, XT, 3, 1, 1, 3, lane);    // W_31
    }
    __syncthreads();
    POTF2_WG_MARK(1);  // (trace builds: the four panels and everything under them | the last step)
    POTF2_WG_MARK(2);
    POTF2_WG_MARK(3);
    // ---- last: X_3b on waves 1..3 | the pivot test and 1 / L_jj on wave 0 ----------------------------------------------------------
    if (wv == 0) {
        // A failed pivot leaves its own diagonal entry NaN or non-positive (d <= 0 or NaN -> r NaN or inf -> d r NaN) and every entry
        // before it a valid positive square root: the first lane that is not > 0 is dpotf2's info.
        const T ljj = S[lane * SLD + lane];
        const unsigned long long bad = __ballot(!(ljj > T(0)));
        const int fail = bad ? (int)__builtin_ctzll(bad) + 1 : 0;
        if (fail == 0) invdiag[lane] = rcp_newton2<T>(ljj);
        if (lane == 0) {
            s_fail = fail;
            // chain.hip: tasks already in flight when a pivot fails run on with garbage, and a LATER diagonal tile may fail on it: the first
            // failure (diagonal tiles finish in order: the smallest pivot) must stay — dpotrf's info
            if (fail) atomicCAS(info, 0, (int)(pivot_base + fail));
        }
    } else {
        const int bb = wv == 1 ? 2 : (wv == 2 ? 0 : 1);
        potf2_x_from_w<T>(w, XT, DI, WTw, 3, bb, lane);
    }
    __syncthreads();
    POTF2_WG_MARK(4);
#undef POTF2_WG_MARK
    return s_fail;
}

}  // namespace
}  // namespace gpmi
