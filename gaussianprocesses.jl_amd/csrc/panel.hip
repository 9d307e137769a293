// panel.hip — the latency-bound pieces around the MFMA update:
//   diag64     64 x 64 diagonal-block Cholesky + explicit inverse, ONE wavefront
//              (the POTF2 of the blocked dpotrf behind make_posdef!, src/GP.jl:110)
//   rows64     one 64-column panel step for the rows below: left-looking update, TRSM as a product with the stored inverse
//              (+ optional refinement step), diagonal-block updates (panel solve; also whiten!, src/GP.jl:27)
//   rows256    the whole 256-column panel step for rows below the diagonal block, in one launch
//   linv256    explicit inverses of the 256 x 256 diagonal blocks; bsolve256 / bsolve_step: backward substitution
//              L' alpha = z  (second half of cK \ y, src/GPE.jl:208)
//   finalize   logdet (PDMats: 2 sum log U_ii) + y'alpha + mll (src/GPE.jl:210)
//   row_gemv / row_var / row_sumsq   predictive mean / variance reductions (src/GP.jl:26,75), diag(K^-1)
#include "common.h"
#include "mfma.h"
#include "potf2.h"

namespace gpmi {

namespace {

template <typename T>
__global__ __launch_bounds__(256, 2) void diag64_kernel(T* __restrict__ A, int64_t ld, T* __restrict__ Linv,
                                                    T* __restrict__ invdiag, int* __restrict__ info, int64_t pivot_base) {
    // Launched with 256 threads of which only wave 0 works: __launch_bounds__(256, 2) is what caps the kernel at 256
    // registers (with a 64-thread bound the 77 KiB of LDS already limits occupancy and the compiler takes 282), and
    // under look-ahead the wave must fit beside a 248-register GEMM wave on its SIMD.
    if (threadIdx.x >= 64) return;
    if (*info != 0) return;
    // Under look-ahead this wave shares a SIMD with two GEMM waves that always have an MFMA ready; instruction
    // arbitration is oldest-first, so without a raised priority the chain crawls (measured: 2 ms instead of 25 us).
    __builtin_amdgcn_s_setprio(3);
    __shared__ T pool[PANEL_POOL];
    diag64_body<T>(A, ld, Linv, invdiag, info, pivot_base, pool);
}

// rows64: the panel step for every row below a 64-wide column block j of a 256-wide panel, 64 rows per workgroup.
//   phase 1 (left-looking)  T = X_j - sum_{s<j} X_s L_js'          K1 = j0 - k0 columns of the same rows
//   phase 2                 X_j <- T * Linv_j'                      the TRSM as a K = 64 product
//   phase 3 (only rows that belong to the panel's own diagonal region; Cholesky)
//                           A_mm -= X_j X_j'                        keeps the next diagonal blocks current
// All products run on the matrix cores from LDS-resident 64 x 64 operands.  Xp / Lp point at COLUMN k0.
template <typename T>
__device__ __forceinline__ void rows64_block(T* __restrict__ Xp, int64_t ldx, int64_t M, int K1, const T* __restrict__ Lp, int64_t ldl,
                                             const T* __restrict__ Linv, int64_t diag_rows, int refine, int64_t row0,
                                             T* __restrict__ buf1, T* __restrict__ buf2) {
    using MF = Mfma<T>;
    using Acc = typename MF::Acc;
    constexpr int LD = 65;
    constexpr int KS = 32, KLD = 33;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;

    Acc acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc_zero<T>(acc[mi][ni]);

    // ---- phase 1 --------------------------------------------------------------------------------------------
    {
        const int r = tid >> 2, kq = (tid & 3) * 8;
        int64_t gr = row0 + r;
        gr = gr < M ? gr : M - 1;
        const T* ga = Xp + gr * ldx + kq;
        const T* gb = Lp + (int64_t)r * ldl + kq;
        T ra[8], rb[8];
        if (K1 > 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                ra[q] = ga[q];
                rb[q] = gb[q];
            }
        }
        for (int ks = 0; ks < K1; ks += KS) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                buf1[r * KLD + kq + q] = ra[q];
                buf2[r * KLD + kq + q] = rb[q];
            }
            __syncthreads();
            if (ks + KS < K1) {  // next slab in flight while this one is multiplied
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    ra[q] = ga[ks + KS + q];
                    rb[q] = gb[ks + KS + q];
                }
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    mma16_nt<T>(acc[mi][ni], buf1 + (wm * 32 + mi * 16) * KLD, KLD, buf2 + (wn * 32 + ni * 16) * KLD, KLD, KS, lane);
            __syncthreads();
        }
    }
    // ---- phase 2 --------------------------------------------------------------------------------------------
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                int64_t gr = row0 + row;
                gr = gr < M ? gr : M - 1;
                buf1[row * LD + col] = Xp[gr * ldx + K1 + col] - acc_get<T>(acc[mi][ni], r);
            }
    for (int e = tid; e < 64 * 64; e += 256) buf2[(e >> 6) * LD + (e & 63)] = Linv[e];
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            acc_zero<T>(acc[mi][ni]);
            mma16_nt<T>(acc[mi][ni], buf1 + (wm * 32 + mi * 16) * LD, LD, buf2 + (wn * 32 + ni * 16) * LD, LD, 64, lane);
        }
    __syncthreads();
    if (refine) {
        // One step of iterative refinement against the block itself:  X <- X + (T - X L_jj') Linv_j'.
        // The product with the explicit inverse is only accurate to cond(L_jj) eps; for covariance blocks whose only
        // regularisation is a 1e-10 nugget (FITC's Kuu, make_posdef!) that is as large as the nugget and the next
        // pivot goes negative where LAPACK's substitution succeeds.  The refined solve has the substitution's accuracy.
        Acc tacc[2][2], xacc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                xacc[mi][ni] = acc[mi][ni];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                    tacc[mi][ni].v[r] = buf1[row * LD + col];  // T, in the accumulator layout
                }
            }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                    buf1[row * LD + col] = acc_get<T>(xacc[mi][ni], r);
                }
        for (int e = tid; e < 64 * 64; e += 256) buf2[(e >> 6) * LD + (e & 63)] = Lp[(int64_t)(e >> 6) * ldl + K1 + (e & 63)];
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc_zero<T>(acc[mi][ni]);
                mma16_nt<T>(acc[mi][ni], buf1 + (wm * 32 + mi * 16) * LD, LD, buf2 + (wn * 32 + ni * 16) * LD, LD, 64, lane);
            }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                    buf1[row * LD + col] = tacc[mi][ni].v[r] - acc_get<T>(acc[mi][ni], r);  // residual
                }
        for (int e = tid; e < 64 * 64; e += 256) buf2[(e >> 6) * LD + (e & 63)] = Linv[e];
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc[mi][ni] = xacc[mi][ni];
                mma16_nt<T>(acc[mi][ni], buf1 + (wm * 32 + mi * 16) * LD, LD, buf2 + (wn * 32 + ni * 16) * LD, LD, 64, lane);
            }
        __syncthreads();
    }
    const bool diag = row0 < diag_rows;  // workgroup-uniform
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                const T v = acc_get<T>(acc[mi][ni], r);
                if (row0 + row < M) Xp[(row0 + row) * ldx + K1 + col] = v;
                if (diag) buf1[row * LD + col] = v;
            }
    // ---- phase 3 --------------------------------------------------------------------------------------------
    if (diag) {
        __syncthreads();
        T* Dg = Xp + row0 * ldx + (K1 + 64 + row0);  // the 64 x 64 diagonal block of these rows
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc_zero<T>(acc[mi][ni]);
                mma16_nt<T>(acc[mi][ni], buf1 + (wm * 32 + mi * 16) * LD, LD, buf1 + (wn * 32 + ni * 16) * LD, LD, 64, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                    Dg[(int64_t)row * ldx + col] -= acc_get<T>(acc[mi][ni], r);
                }
            }
    }
    __syncthreads();  // the next row block reuses buf1 / buf2
}

template <typename T>
__global__ __launch_bounds__(256, 2) void rows64_kernel(T* __restrict__ Xp, int64_t ldx, int64_t M, int K1,
                                                     const T* __restrict__ Lp, int64_t ldl, const T* __restrict__ Linv,
                                                     int64_t diag_rows, const int* __restrict__ info, int refine) {
    if (info && *info != 0) return;
    __builtin_amdgcn_s_setprio(3);  // see diag64_kernel
    __shared__ T pool[2 * 64 * 65];
    // one 64-row block per workgroup, or — when the launch is capped to the slots the persistent update leaves free
    // (chol.h: beside_update) — a grid-stride walk over the row blocks
    const int64_t nblk = (M + 63) / 64;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x)
        rows64_block<T>(Xp, ldx, M, K1, Lp, ldl, Linv, diag_rows, refine, blk * 64, pool, pool + 64 * 65);
}

// ---------------------------------------------------------------------------------------------
// rows256: the whole NB-wide panel step for 64 rows that lie BELOW the panel's diagonal block, in one launch:
//   for j = 0 .. nsub-1:   X_j <- X_j Linv_j' ;   X_i -= X_j L_ij'  (i > j)          (right-looking inside the workgroup)
// The four 64 x 64 blocks of the rows stay in the accumulators (64 fp64 per lane); only the block being applied and one
// 64 x 64 operand (Linv_j or L_ij) are in LDS.  Same arithmetic as four rows64 launches (left-looking there), without
// three launch latencies and without re-reading X: 84 -> ~45 us per panel on the C2 shape.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 1) void rows256_kernel(T* __restrict__ Xp, int64_t ldx, int64_t M, int nsub,
                                                      const T* __restrict__ Lp, int64_t ldl, const T* __restrict__ Linv,
                                                      const int* __restrict__ info) {
    if (info && *info != 0) return;
    using MF = Mfma<T>;
    using Acc = typename MF::Acc;
    constexpr int LD = 65;
    __shared__ T bufX[64 * LD];
    __shared__ T bufL[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int64_t row0 = (int64_t)blockIdx.x * 64;

    // the ten operand blocks are consumed in a fixed order; the next one is always in flight (registers) while the
    // current product runs, and goes to LDS once the product has released the buffer
    T pre[16];
    auto fetch = [&](const T* src, int64_t ld) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q;
            pre[q] = src[(int64_t)(e >> 6) * ld + (e & 63)];
        }
    };
    auto publish = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = tid + 256 * q;
            bufL[(e >> 6) * LD + (e & 63)] = pre[q];
        }
    };
    fetch(Linv, 64);

    Acc acc[4][2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                    int64_t gr = row0 + row;
                    gr = gr < M ? gr : M - 1;
                    acc[j][mi][ni].v[r] = (j < nsub) ? Xp[gr * ldx + j * 64 + col] : T(0);
                }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nsub) {  // workgroup-uniform
            // X_j <- X_j Linv_j'
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                        bufX[row * LD + col] = acc[j][mi][ni].v[r];
                    }
            publish();  // Linv_j
            if (j + 1 < nsub) fetch(Lp + (int64_t)((j + 1) * 64) * ldl + j * 64, ldl);  // L_{j+1,j}
            __syncthreads();
            Acc w[2][2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc_zero<T>(w[mi][ni]);
                    mma16_nt<T>(w[mi][ni], bufX + (wm * 32 + mi * 16) * LD, LD, bufL + (wn * 32 + ni * 16) * LD, LD, 64, lane);
                }
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                        const T v = acc_get<T>(w[mi][ni], r);
                        if (row0 + row < M) Xp[(row0 + row) * ldx + j * 64 + col] = v;
                        bufX[row * LD + col] = -v;  // the updates below ADD (-X_j) L_ij'
                    }
#pragma unroll
            for (int i = j + 1; i < 4; ++i) {
                if (i < nsub) {
                    publish();  // L_ij
                    if (i + 1 < nsub) fetch(Lp + (int64_t)((i + 1) * 64) * ldl + j * 64, ldl);   // L_{i+1,j}
                    else fetch(Linv + (int64_t)(j + 1) * 64 * 64, 64);                           // Linv_{j+1}  (j + 1 <= i < nsub)
                    __syncthreads();
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            mma16_nt<T>(acc[i][mi][ni], bufX + (wm * 32 + mi * 16) * LD, LD, bufL + (wn * 32 + ni * 16) * LD, LD, 64, lane);
                    __syncthreads();
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bsolve_step: one 64-block of the backward solve  L' alpha = z.  With the stored inverse of the diagonal block
// the block solve is a 64 x 64 mat-vec  alpha_b = Linv_b' z_b  (every workgroup recomputes it — cheaper than an
// extra launch on the critical path), workgroup 0 publishes alpha_b, then all workgroups apply
// z[j] -= sum_i L[j0+i][j] * alpha_b[i]  to their 256 columns (coalesced rows of L).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bsolve_step_kernel(const T* __restrict__ Arow, int64_t ld, int64_t j0,
                                                          const T* __restrict__ Linv, T* __restrict__ z,
                                                          T* __restrict__ alpha) {
    __shared__ T SLI[64 * 65];
    __shared__ T sz[64];
    __shared__ T part[4][64];
    __shared__ T sal[64];
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * 64; e += 256) SLI[(e >> 6) * 65 + (e & 63)] = Linv[e];
    if (tid < 64) sz[tid] = z[j0 + tid];
    __syncthreads();
    {   // alpha_i = sum_t Linv[t][i] z_t : 4 partial sums over t (one per wavefront)
        const int i = tid & 63, q = tid >> 6;
        T s = T(0);
#pragma unroll
        for (int t = q * 16; t < q * 16 + 16; ++t) s += SLI[t * 65 + i] * sz[t];
        part[q][i] = s;
    }
    __syncthreads();
    if (tid < 64) {
        const T a = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        sal[tid] = a;
        if (blockIdx.x == 0) alpha[j0 + tid] = a;
    }
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * 256 + tid;
    if (j < j0) {
        const T* col = Arow + j;  // Arow = row j0 of the factor
        T acc = T(0);
#pragma unroll 8
        for (int i = 0; i < 64; ++i) acc += col[(int64_t)i * ld] * sal[i];
        z[j] -= acc;
    }
}

// ---------------------------------------------------------------------------------------------
// linv256: explicit inverse of every NB x NB diagonal block of the factor, assembled from the stored 64 x 64 inverses.
// Workgroup (b, j) produces column block j of block b's inverse Out = L_bb^-1.  With Wt = Out' this is the whitening
// recurrence applied to rows j of an identity:
//     Wt[j][j] = Linv_j' ,   Wt[j][m] = -( sum_{t=j}^{m-1} Wt[j][t] L[m][t]' ) Linv_m'      (m > j)
// run right-looking: as soon as Wt[j][t] exists its contribution to every later sum is accumulated (registers), so only
// the current block lives in LDS.  All products are 64 x 64 x 64 on the matrix cores.  With the NB-inverse every later
// triangular solve against the factor (predict whitening, back-substitution) is ONE product per NB columns.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void linv256_block(int vb, T* __restrict__ bufW, T* __restrict__ bufL, const T* __restrict__ A, int64_t ld,
                                              const T* __restrict__ linv64, T* __restrict__ out, int64_t npad) {
    using MF = Mfma<T>;
    using Acc = typename MF::Acc;
    constexpr int LD = 65;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int b = vb >> 2, j = vb & 3;
    const int64_t k0 = (int64_t)b * NB;
    const int64_t rem = npad - k0;
    const int nb = (int)((rem < NB ? rem : NB) / 64);
    if (j >= nb) return;
    T* O = out + (int64_t)b * NB * NB;  // row-major, leading dimension NB
    const T* Lj = linv64 + (k0 / 64 + j) * 64 * 64;
    for (int i = 0; i < j; ++i)
        for (int e = tid; e < 64 * 64; e += 256) O[(int64_t)(i * 64 + (e >> 6)) * NB + j * 64 + (e & 63)] = T(0);
    for (int e = tid; e < 64 * 64; e += 256) {
        const int r = e >> 6, cc = e & 63;
        const T v = Lj[e];
        O[(int64_t)(j * 64 + r) * NB + j * 64 + cc] = v;
        bufW[cc * LD + r] = v;  // Wt[j][j] = Linv_j'
    }
    Acc acc[3][2][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc_zero<T>(acc[t][mi][ni]);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m >= j && m < nb) {  // workgroup-uniform
            if (m > j) {
                {
                    // Wt[j][m] = -(acc_m) Linv_m'
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                                bufW[row * LD + col] = acc_get<T>(acc[m > 0 ? m - 1 : 0][mi][ni], r);
                            }
                    const T* Lm = linv64 + (k0 / 64 + m) * 64 * 64;
                    for (int e = tid; e < 64 * 64; e += 256) bufL[(e >> 6) * LD + (e & 63)] = Lm[e];
                    __syncthreads();
                    Acc w[2][2];
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            acc_zero<T>(w[mi][ni]);
                            mma16_nt<T>(w[mi][ni], bufW + (wm * 32 + mi * 16) * LD, LD, bufL + (wn * 32 + ni * 16) * LD, LD, 64, lane);
                        }
                    __syncthreads();
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = wm * 32 + mi * 16 + MF::row_of(lane, r), col = wn * 32 + ni * 16 + MF::col_of(lane, r);
                                const T v = -acc_get<T>(w[mi][ni], r);
                                bufW[row * LD + col] = v;                                          // Wt[j][m]
                                O[(int64_t)(m * 64 + col) * NB + j * 64 + row] = v;               // Out[m][j] = Wt[j][m]'
                            }
                    __syncthreads();
                }
            }
#pragma unroll
            for (int i = m + 1; i < 4; ++i) {
                if (i < nb) {
                    const T* Lim = A + (k0 + i * 64) * ld + k0 + m * 64;
                    for (int e = tid; e < 64 * 64; e += 256) bufL[(e >> 6) * LD + (e & 63)] = Lim[(int64_t)(e >> 6) * ld + (e & 63)];
                    __syncthreads();
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            mma16_nt<T>(acc[i - 1][mi][ni], bufW + (wm * 32 + mi * 16) * LD, LD, bufL + (wn * 32 + ni * 16) * LD, LD, 64, lane);
                    __syncthreads();
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void linv256_kernel(const T* __restrict__ A, int64_t ld, const T* __restrict__ linv64,
                                                      T* __restrict__ out, int64_t npad, const int* __restrict__ info) {
    if (info && *info != 0) return;
    __shared__ T bufW[64 * 65];
    __shared__ T bufL[64 * 65];
    const int nvb = 4 * (int)((npad + NB - 1) / NB);  // (block, column block) pairs; grid-stride when the launch is capped
    for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        linv256_block<T>(vb, bufW, bufL, A, ld, linv64, out, npad);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// place_inv_blocks: level 0 of the super-panel inverse (chol.h build_super_inverse).  The packed NB x NB inverses (linv256
// layout) go onto the diagonal of the w x w matrix LW (leading dimension wld) and, transposed, of LWT; every other 64 x 64
// tile of both is zeroed (the merge products accumulate into them, and the triangular product reads the zeros above the
// diagonal of its diagonal tiles).  Grid-stride over the (w/64)^2 tiles: the launch may be capped to a few workgroups.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void place_inv_blocks_kernel(const T* __restrict__ l256, T* __restrict__ LW, T* __restrict__ LWT,
                                                               int64_t wld, int w64) {
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < w64 * w64; t += gridDim.x) {
        const int gi = t / w64, gj = t % w64;  // 64 x 64 tile (gi, gj) of the w x w matrix
        if ((gi >> 2) != (gj >> 2)) {
            for (int r = ty; r < 64; r += 4) {
                LW[(int64_t)(gi * 64 + r) * wld + gj * 64 + tx] = T(0);
                LWT[(int64_t)(gi * 64 + r) * wld + gj * 64 + tx] = T(0);
            }
            continue;
        }
        const int b = gi >> 2, ti = gi & 3, tj = gj & 3;
        const T* src = l256 + (int64_t)b * NB * NB;
        for (int r = ty; r < 64; r += 4) {
            const T v = src[(int64_t)(ti * 64 + r) * NB + tj * 64 + tx];
            LW[(int64_t)(gi * 64 + r) * wld + gj * 64 + tx] = v;
            tile[r][tx] = v;
        }
        __syncthreads();
        // LWT tile (gj, gi) = transpose of LW tile (gi, gj); every diagonal-block tile position is written exactly once
        for (int r = ty; r < 64; r += 4) LWT[(int64_t)(gj * 64 + r) * wld + gi * 64 + tx] = tile[tx][r];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// bsolve256: one NB-block of the backward solve  L' alpha = z  through the block's explicit inverse:
//   alpha_b = Linv_b' z_b   (every workgroup recomputes it: column-coalesced reads of the 512 KiB inverse from L2),
//   z[j] -= sum_i L[k0+i][j] alpha_b[i]   on the workgroup's 256 columns left of the block.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void bsolve256_kernel(const T* __restrict__ Arow, int64_t ld, int64_t k0, int nbk,
                                                         const T* __restrict__ Linv, int64_t ldi, T* __restrict__ z, T* __restrict__ alpha) {
    __shared__ T sz[NB];
    __shared__ T sal[NB];
    __shared__ T part[4][NB];
    const int tid = threadIdx.x, c = tid & (NB - 1), q = tid >> 8;  // column c, row quarter q (64 rows, 16 loads in flight)
    if (tid < NB) sz[tid] = tid < nbk ? z[k0 + tid] : T(0);
    __syncthreads();
    const int r_lo = q * 64, r_hi = (r_lo + 64 < nbk) ? r_lo + 64 : nbk;
    {
        T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
        if (c < nbk && r_hi > c) {  // Linv[r][c] = 0 for r < c: quarters entirely above the diagonal are skipped
            const T* col = Linv + c;
#pragma unroll 4
            for (int r = r_lo; r < r_hi; r += 4) {
                s0 += col[(int64_t)r * ldi] * sz[r];
                s1 += col[(int64_t)(r + 1) * ldi] * sz[r + 1];
                s2 += col[(int64_t)(r + 2) * ldi] * sz[r + 2];
                s3 += col[(int64_t)(r + 3) * ldi] * sz[r + 3];
            }
        }
        part[q][c] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < NB) {
        const T a = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
        sal[tid] = a;
        if (blockIdx.x == 0 && tid < nbk) alpha[k0 + tid] = a;
    }
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * NB + c;
    {
        T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
        if (j < k0) {
            const T* col = Arow + j;  // Arow = row k0 of the factor
#pragma unroll 4
            for (int i = r_lo; i < r_hi; i += 4) {
                a0 += col[(int64_t)i * ld] * sal[i];
                a1 += col[(int64_t)(i + 1) * ld] * sal[i + 1];
                a2 += col[(int64_t)(i + 2) * ld] * sal[i + 2];
                a3 += col[(int64_t)(i + 3) * ld] * sal[i + 3];
            }
        }
        part[q][c] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (tid < NB && j < k0) z[j] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
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

// out[p] = sum_{j >= j0(p)} R[p][j]^2,  j0(p) = (p / blk) * blk for blk > 0 (rows of a block upper-triangular matrix whose
// entries left of the row's own block were never written), 0 otherwise
template <typename T>
__global__ __launch_bounds__(256) void row_sumsq_kernel(const T* __restrict__ R, int64_t ldr, int64_t n, int64_t blk,
                                                        T* __restrict__ out) {
    __shared__ double sh[256];
    const T* row = R + (int64_t)blockIdx.x * ldr;
    double s = 0.0;
    const int64_t j0 = blk > 0 ? ((int64_t)blockIdx.x / blk) * blk : 0;
    for (int64_t j = j0 + threadIdx.x; j < n; j += 256) {
        const double v = (double)row[j];
        s += v * v;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (T)sh[0];
}

}  // namespace

template <typename T>
void launch_place_inv_blocks(gpmi_ctx* ctx, const T* l256, T* LW, T* LWT, int64_t wld, int nblk) {
    const int w64 = nblk * (NB / 64);
    hipLaunchKernelGGL(place_inv_blocks_kernel<T>, dim3((unsigned)side_cap(ctx, (int64_t)w64 * w64)), dim3(256), 0, ctx->stream, l256, LW, LWT,
                       wld, w64);
}

template <typename T>
void launch_diag64(gpmi_ctx* ctx, T* A, int64_t ld, T* linv, T* invdiag, int* info, int64_t pivot_base) {
    ProfScope ps(ctx, GPMI_PROF_PANEL, 2.0 * 64.0 * 64.0 * 64.0 / 3.0, 0.0, false, /*chain_kernel=*/true);
    hipLaunchKernelGGL(diag64_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, A, ld, linv, invdiag, info, pivot_base);
}
template <typename T>
void launch_rows64(gpmi_ctx* ctx, T* Xp, int64_t ldx, int64_t M, int K1, const T* Lp, int64_t ldl, const T* linv,
                   int64_t diag_rows, const int* info) {
    if (M <= 0) return;
    ProfScope ps(ctx, GPMI_PROF_PANEL, 2.0 * (double)M * 64.0 * (double)(K1 + 64), 0.0, false, /*chain_kernel=*/true);
    hipLaunchKernelGGL(rows64_kernel<T>, dim3((unsigned)side_cap(ctx, (M + 63) / 64)), dim3(256), 0, ctx->stream, Xp, ldx, M, K1, Lp, ldl,
                       linv, diag_rows, info, ctx->refine_solves ? 1 : 0);
}
template <typename T>
void launch_rows256(gpmi_ctx* ctx, T* Xp, int64_t ldx, int64_t M, int nsub, const T* Lp, int64_t ldl, const T* linv,
                    const int* info) {
    if (M <= 0) return;
    ProfScope ps(ctx, GPMI_PROF_PANEL, 2.0 * (double)M * 64.0 * 64.0 * (double)(nsub * (nsub + 1) / 2), 0.0, false, /*chain_kernel=*/true);
    hipLaunchKernelGGL(rows256_kernel<T>, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, ctx->stream, Xp, ldx, M, nsub, Lp, ldl,
                       linv, info);
}
template <typename T>
void launch_bsolve_step(gpmi_ctx* ctx, const T* Arow, int64_t ld, int64_t j0, const T* linv, T* z, T* alpha) {
    const unsigned blocks = (unsigned)(j0 > 0 ? (j0 + 255) / 256 : 1);
    hipLaunchKernelGGL(bsolve_step_kernel<T>, dim3(blocks), dim3(256), 0, ctx->stream, Arow, ld, j0, linv, z, alpha);
}
template <typename T>
void launch_linv256(gpmi_ctx* ctx, const T* A, int64_t ld, const T* linv64, T* out, int64_t npad, const int* info) {
    const unsigned nblk = (unsigned)((npad + NB - 1) / NB);
    hipLaunchKernelGGL(linv256_kernel<T>, dim3((unsigned)side_cap(ctx, 4 * nblk)), dim3(256), 0, ctx->stream, A, ld, linv64, out, npad, info);
}
template <typename T>
void launch_bsolve256(gpmi_ctx* ctx, const T* Arow, int64_t ld, int64_t k0, int nbk, const T* linv256, T* z, T* alpha, int64_t ldinv) {
    const unsigned blocks = (unsigned)(k0 > 0 ? (k0 + 255) / 256 : 1);
    hipLaunchKernelGGL(bsolve256_kernel<T>, dim3(blocks), dim3(1024), 0, ctx->stream, Arow, ld, k0, nbk, linv256, ldinv, z, alpha);
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
void launch_row_sumsq(gpmi_ctx* ctx, const T* R, int64_t ldr, int64_t P, int64_t n, int64_t blk, T* out) {
    if (P <= 0) return;
    hipLaunchKernelGGL(row_sumsq_kernel<T>, dim3((unsigned)P), dim3(256), 0, ctx->stream, R, ldr, n, blk, out);
}

template <typename T>
void launch_logdiag(gpmi_ctx* ctx, const T* A, int64_t ld, int64_t nrows, int64_t col_off, double* out) {
    hipLaunchKernelGGL(logdiag_kernel<T>, dim3(1), dim3(1024), 0, ctx->stream, A, ld, nrows, col_off, out);
}

#define INST(T)                                                                                                   \
    template void launch_diag64<T>(gpmi_ctx*, T*, int64_t, T*, T*, int*, int64_t);                                \
    template void launch_rows64<T>(gpmi_ctx*, T*, int64_t, int64_t, int, const T*, int64_t, const T*, int64_t,    \
                                   const int*);                                                                   \
    template void launch_rows256<T>(gpmi_ctx*, T*, int64_t, int64_t, int, const T*, int64_t, const T*, const int*); \
    template void launch_bsolve_step<T>(gpmi_ctx*, const T*, int64_t, int64_t, const T*, T*, T*);                 \
    template void launch_linv256<T>(gpmi_ctx*, const T*, int64_t, const T*, T*, int64_t, const int*);             \
    template void launch_place_inv_blocks<T>(gpmi_ctx*, const T*, T*, T*, int64_t, int);             \
    template void launch_bsolve256<T>(gpmi_ctx*, const T*, int64_t, int64_t, int, const T*, T*, T*, int64_t);     \
    template void launch_finalize<T>(gpmi_ctx*, const T*, int64_t, int64_t, const T*, const T*, double*);         \
    template void launch_row_gemv<T>(gpmi_ctx*, const T*, int64_t, int64_t, int64_t, const T*, const T*, T*);     \
    template void launch_row_var<T>(gpmi_ctx*, const T*, int64_t, int64_t, int64_t, double, T*);                  \
    template void launch_row_sumsq<T>(gpmi_ctx*, const T*, int64_t, int64_t, int64_t, int64_t, T*);               \
    template void launch_logdiag<T>(gpmi_ctx*, const T*, int64_t, int64_t, int64_t, double*);
INST(double)
INST(float)

}  // namespace gpmi
