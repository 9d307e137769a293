// grad.hip — gradient of the marginal log-likelihood with respect to the kernel hyper-parameters
// (update_dmll!, src/GPE.jl:298-324; dmll_kern!, src/GPE.jl:219-241; dmll_noise, src/GPE.jl:273-275).
//
// The reference materialises W = alpha alpha' - K^-1 (get_ααinvcKI!, 2 N^3 flop through dpotrs on the identity) and
// then runs a single-threaded double loop calling dKij_dθ! per entry.  Here:
//   * K^-1 = L^-T L^-1 is built from rows of L^-T (the whiten machinery applied to the identity, growing row count)
//     and one MFMA product that starts its K loop at the tile's first row — N^3/3 + N^3/3 flop (api.hip);
//   * dmll_kernel regenerates dK_ij/dθ_p for ALL parameters of the kernel tree on the fly from LDS-staged x blocks
//     and reduces  sum_{i>=j} W_ij dK_ij/dθ_p (diagonal counted half) without ever forming an N x N x p stack:
//     per leaf one forward-mode sweep of the postfix program (product rule of prod_kernel.jl:17-68, sum rule of
//     sum_kernel.jl:18-51) on a register stack, then that leaf's closed-form derivatives
//     (se_iso.jl:41-50, se_ard.jl:45-54, mat.jl:5-33 + mat*_*.jl, rq_iso.jl:45-61, rq_ard.jl:48-63, noise.jl:47,
//     const.jl:40).  Per-thread accumulators live in LDS ([slot][thread], conflict-free); block partials are
//     summed by a second kernel in a fixed order, so the gradient is bit-reproducible.
// The trace of W (noise gradient) rides along as one more slot.
#include "common.h"

namespace gpmi {

namespace {

template <typename T>
__global__ void set_identity_kernel(T* __restrict__ A, int64_t ld, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i * ld + i] = T(1);
}

template <typename T>
struct GM;
template <>
struct GM<double> {
    static __device__ __forceinline__ double exp_(double x) { return exp(x); }
    static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
    static __device__ __forceinline__ double pow_(double x, double y) { return pow(x, y); }
    static __device__ __forceinline__ double log_(double x) { return log(x); }
    static constexpr double rtol = 1.4901161193847656e-08;
};
template <>
struct GM<float> {
    static __device__ __forceinline__ float exp_(float x) { return expf(x); }
    static __device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float pow_(float x, float y) { return powf(x, y); }
    static __device__ __forceinline__ float log_(float x) { return logf(x); }
    static constexpr float rtol = 3.4526698300124393e-04f;
};

__device__ __forceinline__ bool is_ard_op(int op) {
    return op == GPMI_K_SE_ARD || op == GPMI_K_MAT12_ARD || op == GPMI_K_MAT32_ARD || op == GPMI_K_MAT52_ARD || op == GPMI_K_RQ_ARD;
}

// RECT (FITC gradient, fitc.hip): the same reduction over a RECTANGLE of pairs (x_i, xb_j), i < n, j < nb, with an explicit
// weight matrix:  partial[p] = sum_ij Wt[i][j] dk(x_i, xb_j)/dθ_p  — `Kinv` is Wt, alpha is unused, no triangle, no trace slot.
// GEN (DMAX == 0): the form without limits — any input dimension, any number of hyper-parameters (the reference's dmll_kern! loops
// over whatever the kernel has, src/GPE.jl:219-241).  Nothing per-dimension or per-parameter lives in registers or in a
// [slot][thread] LDS table: the pair's differences are re-read from global memory (L1 / L2) wherever they are used, and every
// contribution is reduced across the wave at once (shuffles: a fixed order, so still bit-reproducible) into a [wave][slot] table.
// Several times slower per pair than the register forms; it only runs for d > 32 or more than 64 hyper-parameters.
template <typename T, int DMAX, bool RECT>
__global__ __launch_bounds__(256) void dmll_kernel(const T* __restrict__ x, int64_t n, int d, const T* __restrict__ alpha,
                                                   const T* __restrict__ Kinv, int64_t ld,
                                                   const DevProgram* __restrict__ prog, double* __restrict__ partial,
                                                   int n_hyp, const T* __restrict__ xb_pts, int64_t nb) {
    constexpr int GSTK = 6;  // evaluation-stack depth (validated on the host, as for cov)
    constexpr bool GEN = DMAX == 0;
    constexpr int DREG = GEN ? 1 : DMAX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* gl = reinterpret_cast<double*>(smem);                                   // [n_hyp + 1][256] accumulators (GEN: none)
    T* sa = reinterpret_cast<T*>(gl + (GEN ? 0 : (size_t)(n_hyp + 1) * 256));       // [64][d] row points (GEN: none)
    T* sal = sa + (GEN ? 0 : 64 * d);                                               // [64] alpha of the rows
    double* red = reinterpret_cast<double*>(sal + 64);                              // [4][n_hyp + 1]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.y * 64, col0 = (int64_t)blockIdx.x * 64;
    const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int nslots = n_hyp + 1;
    if ((!RECT && col0 > row0 + 63) || row0 >= n || (RECT && col0 >= nb)) {  // nothing on or below the diagonal in this tile
        for (int s = tid; s < nslots; s += 256) partial[bid * nslots + s] = 0.0;
        return;
    }
    if constexpr (GEN) {
        for (int s = tid; s < 4 * nslots; s += 256) red[s] = 0.0;
    } else {
        for (int s = 0; s < nslots; ++s) gl[s * 256 + tid] = 0.0;
        for (int e = tid; e < 64 * d; e += 256) {
            int r = e / d, k = e - r * d;
            int64_t gr = row0 + r;
            gr = gr < n ? gr : n - 1;
            sa[e] = x[gr * d + k];
        }
    }
    if (tid < 64) {
        int64_t gr = row0 + tid;
        sal[tid] = (!RECT && gr < n) ? alpha[gr] : T(0);
    }
    const int64_t gcol = col0 + lane;
    const int64_t ncols = RECT ? nb : n;
    const T* __restrict__ xcols = RECT ? xb_pts : x;
    const int64_t gc = gcol < ncols ? gcol : ncols - 1;
    T xb[DREG];
#pragma unroll
    for (int k = 0; k < DREG; ++k) xb[k] = (!GEN && k < d) ? xcols[gc * d + k] : T(0);
    const T* __restrict__ xc = xcols + gc * d;  // GEN: this lane's column point, read where it is used
    const T acol = (!RECT && gcol < n) ? alpha[gcol] : T(0);
    __syncthreads();
    // accumulate `v` into hyper-parameter slot `slot` (every call site is wave-uniform)
    auto acc = [&](int slot, double v) __attribute__((always_inline)) {
        if constexpr (GEN) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[wv * nslots + slot] += v;
        } else {
            gl[slot * 256 + tid] += v;
        }
    };

    const int nops = prog->n_ops;
    for (int rr = 0; rr < 16; ++rr) {
        const int row = wv * 16 + rr;
        const int64_t grow = row0 + row;
        if (grow >= n) break;  // wave-uniform
        const bool valid = RECT ? (gcol < nb) : (gcol <= grow);  // lower triangle including the diagonal (gcol < n follows)
        const T* sar = GEN ? (x + grow * d) : (sa + row * d);  // GEN: the row point straight from global memory (wave-uniform address)
        T dsq[DREG];
#pragma unroll
        for (int k = 0; k < DREG; ++k) {
            const T df = (!GEN && k < d) ? (sar[k] - xb[k]) : T(0);
            dsq[k] = df * df;
        }
        // W_ij and its weight: diagonal entries count half (GPE.jl:228-231), strict lower entries once (:233-238)
        const T kin = valid ? Kinv[grow * ld + gcol] : T(0);
        const T wij = RECT ? kin : (nb < 0 ? sal[row] * acol + kin : sal[row] * acol - kin);  // nb < 0: Kinv holds -K^-1
        const T ww = valid ? ((!RECT && grow == gcol) ? T(0.5) * wij : wij) : T(0);
        acc(n_hyp, (!RECT && valid && grow == gcol) ? (double)wij : 0.0);  // tr(alpha alpha' - K^-1)

        // For every leaf L: one forward-mode sweep of the postfix program with the seed on L gives
        //   kv = value of L,  m = d(root)/d(value of L)   (sum rule: sum_kernel.jl:18-51, product rule: prod_kernel.jl:17-68)
        // on a shifting register stack (static indices only — no per-thread arrays, nothing in scratch).  Trees are
        // small, so re-evaluating the other leaves per sweep (n_leaves^2 leaf evaluations) is cheaper than spilling.
        for (int L = 0; L < nops; ++L) {
            const int opL = prog->leaf[L].op;
            if (opL == GPMI_K_SUM || opL == GPMI_K_PROD) continue;
            T sv[GSTK], sd[GSTK];
#pragma unroll
            for (int q = 0; q < GSTK; ++q) {
                sv[q] = T(0);
                sd[q] = T(0);
            }
            T kv = T(0), r2L = T(0);
            for (int o = 0; o < nops; ++o) {
                const int op = prog->leaf[o].op;
                if (op == GPMI_K_SUM || op == GPMI_K_PROD) {
                    const T a = sv[1], da = sd[1], bq = sv[0], db = sd[0];
                    sv[0] = (op == GPMI_K_SUM) ? a + bq : a * bq;
                    sd[0] = (op == GPMI_K_SUM) ? da + db : da * bq + a * db;
#pragma unroll
                    for (int q = 1; q < GSTK - 1; ++q) {
                        sv[q] = sv[q + 1];
                        sd[q] = sd[q + 1];
                    }
                    continue;
                }
                T v, r2 = T(0);
                const T s2 = (T)prog->leaf[o].s2;
                const double* w = prog->wtab() + prog->leaf[o].woff;
                if (op == GPMI_K_CONST) {
                    v = s2;
                } else if (op == GPMI_K_NOISE) {
                    bool same = true;
                    if constexpr (GEN) {
                        for (int k = 0; k < d; ++k) {
                            if (w[k] != 0.0) {
                                const T a = sar[k], b = xc[k];
                                const T m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
                                same = same && ((a == b) || (fabs(a - b) <= GM<T>::rtol * m));
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < DREG; ++k) {
                            if (k < d && w[k] != 0.0) {
                                const T a = sar[k], b = xb[k];
                                const T m = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
                                same = same && ((a == b) || (fabs(a - b) <= GM<T>::rtol * m));
                            }
                        }
                    }
                    v = same ? s2 : T(0);
                } else {
                    if constexpr (GEN) {
                        for (int k = 0; k < d; ++k) {
                            const T df = sar[k] - xc[k];
                            r2 += (df * df) * (T)w[k];
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < DREG; ++k)
                            if (k < d) r2 += dsq[k] * (T)w[k];
                    }
                    const T p0 = (T)prog->leaf[o].p0, al = (T)prog->leaf[o].p1;
                    switch (op) {
                        case GPMI_K_SE_ISO: v = s2 * GM<T>::exp_((T(-0.5) * r2) * p0); break;
                        case GPMI_K_SE_ARD: v = s2 * GM<T>::exp_(T(-0.5) * r2); break;
                        case GPMI_K_MAT12_ISO:
                        case GPMI_K_MAT12_ARD: v = s2 * GM<T>::exp_(-(GM<T>::sqrt_(r2) * p0)); break;
                        case GPMI_K_MAT32_ISO:
                        case GPMI_K_MAT32_ARD: {
                            const T s = T(1.7320508075688772935) * GM<T>::sqrt_(r2) * p0;
                            v = s2 * (T(1) + s) * GM<T>::exp_(-s);
                        } break;
                        case GPMI_K_MAT52_ISO:
                        case GPMI_K_MAT52_ARD: {
                            const T s = T(2.2360679774997896964) * GM<T>::sqrt_(r2) * p0;
                            v = s2 * (T(1) + s + s * s * T(1.0 / 3.0)) * GM<T>::exp_(-s);
                        } break;
                        default: v = s2 * GM<T>::pow_(T(1) + r2 * p0, -al); break;  // RQ iso / ard
                    }
                }
                if (o == L) {
                    kv = v;
                    r2L = r2;
                }
#pragma unroll
                for (int q = GSTK - 1; q > 0; --q) {
                    sv[q] = sv[q - 1];
                    sd[q] = sd[q - 1];
                }
                sv[0] = v;
                sd[0] = (o == L) ? T(1) : T(0);
            }
            const T a = sd[0] * ww;  // weight of this leaf's derivatives in the reduction
            const int op = opL;
            const int poff = prog->leaf[L].poff;
            if (op == GPMI_K_NOISE || op == GPMI_K_CONST) {
                acc(poff, (double)(a * T(2) * kv));  // noise.jl:47-48, const.jl:40
                continue;
            }
            const T r2 = r2L;
            const T s2 = (T)prog->leaf[L].s2, p0 = (T)prog->leaf[L].p0, al = (T)prog->leaf[L].p1;
            const int nd = prog->leaf[L].nd;
            const bool ard = is_ard_op(op);
            const int sig = poff + (ard ? nd : 1);  // slot of log sigma
            acc(sig, (double)(a * T(2) * kv));  // dk_dlσ = 2k (stationary.jl:28)
            // length-scale derivative = c * (r2 for iso | w_k dsq_k for ARD dim k)
            T c;
            const T re = GM<T>::sqrt_(r2);
            switch (op) {
                case GPMI_K_SE_ISO: c = kv * p0; break;                         // r/l2 * k
                case GPMI_K_SE_ARD: c = kv; break;                              // wdiff * k
                case GPMI_K_MAT12_ISO: c = (re > T(0)) ? kv * p0 / re : T(0); break;   // r/l * k  = r2 * (k p0 / r)
                case GPMI_K_MAT12_ARD: c = (re > T(0)) ? kv / re : T(0); break;        // wdiff / r * k
                case GPMI_K_MAT32_ISO: c = T(3) * s2 * p0 * p0 * GM<T>::exp_(-T(1.7320508075688772935) * re * p0); break;  // s2 s^2 e^-s
                case GPMI_K_MAT32_ARD: c = T(3) * s2 * GM<T>::exp_(-T(1.7320508075688772935) * re); break;
                case GPMI_K_MAT52_ISO: {
                    const T s = T(2.2360679774997896964) * re * p0;
                    c = T(5.0 / 3.0) * s2 * p0 * p0 * (T(1) + s) * GM<T>::exp_(-s);   // s2/3 s^2 (1+s) e^-s
                } break;
                case GPMI_K_MAT52_ARD: {
                    const T s = T(2.2360679774997896964) * re;
                    c = T(5.0 / 3.0) * s2 * (T(1) + s) * GM<T>::exp_(-s);
                } break;
                case GPMI_K_RQ_ISO: {  // s2 * s * part^(-a-1), s = r2 / l2 = r2 * p0 * 2a
                    const T part = T(1) + r2 * p0;
                    c = kv / part * (p0 * T(2) * al);
                } break;
                default: {  // RQ_ARD: s2 * wdiff * part^(-a-1)
                    const T part = T(1) + r2 * p0;
                    c = kv / part;
                } break;
            }
            if (ard) {
                const double* w = prog->wtab() + prog->leaf[L].woff;
                const int32_t* pm = prog->pmtab() + prog->leaf[L].woff;
                if constexpr (GEN) {
                    for (int k = 0; k < d; ++k) {
                        const int z = pm[k];
                        if (z >= 0) {
                            const T df = sar[k] - xc[k];
                            acc(poff + z, (double)(a * c * (df * df) * (T)w[k]));
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < DREG; ++k) {
                        if (k < d) {
                            const int z = pm[k];
                            if (z >= 0) acc(poff + z, (double)(a * c * dsq[k] * (T)w[k]));
                        }
                    }
                }
            } else {
                acc(poff, (double)(a * c * r2));
            }
            if (op == GPMI_K_RQ_ISO || op == GPMI_K_RQ_ARD) {  // dk/d(log alpha) = k (s/(2 part) - a log part)
                const T part = T(1) + r2 * p0;
                const T half_s = r2 * p0 * al;  // iso: s/2 = r2/(2 l2) = r2 p0 a ; ard: r/2 = r2 p0 a (p0 = 0.5/a)
                acc(sig + 1, (double)(a * kv * (half_s / part - al * GM<T>::log_(part))));
            }
        }
    }
    // ---- block reduction: wave shuffles, then four partials per slot --------------------------------------------
    __syncthreads();
    if constexpr (!GEN) {
        for (int s = 0; s < nslots; ++s) {
            double v = gl[s * 256 + tid];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[wv * nslots + s] = v;
        }
        __syncthreads();
    }
    for (int s = tid; s < nslots; s += 256)
        partial[bid * nslots + s] = (red[s] + red[nslots + s]) + (red[2 * nslots + s] + red[3 * nslots + s]);
}

__global__ __launch_bounds__(1024) void reduce_partials_kernel(const double* __restrict__ partial, int64_t nblocks, int nslots,
                                                              double* __restrict__ out) {
    __shared__ double sh[1024];
    const int s = blockIdx.x;
    double v = 0.0;
    for (int64_t b = threadIdx.x; b < nblocks; b += 1024) v += partial[b * nslots + s];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[s] = sh[0];
}

}  // namespace

template <typename T>
void launch_set_identity(gpmi_ctx* ctx, T* A, int64_t ld, int64_t n) {
    hipMemsetAsync(A, 0, (size_t)(n * ld) * sizeof(T), ctx->stream);
    hipLaunchKernelGGL(set_identity_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, A, ld, n);
}

template <typename T, bool RECT>
static int64_t launch_dmll_any(gpmi_ctx* ctx, const T* x, int64_t n, int d, const T* alpha, const T* Kinv, int64_t ld, double* partial,
                               int n_hyp, const T* xb, int64_t nb) {
    const unsigned ntr = (unsigned)((n + 63) / 64), ntc = RECT ? (unsigned)((nb + 63) / 64) : ntr;
    const bool gen = d > GRAD_MAX_D || n_hyp > GRAD_MAX_HYP;  // beyond the register / LDS forms: the form without limits
    const size_t lds = gen ? (size_t)64 * sizeof(T) + (size_t)4 * (n_hyp + 1) * 8
                           : (size_t)(n_hyp + 1) * 256 * 8 + (size_t)(64 * d + 64) * sizeof(T) + (size_t)4 * (n_hyp + 1) * 8;
    auto go = [&](auto kern) {
        if (lds > 48 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(ntc, ntr), dim3(256), lds, ctx->stream, x, n, d, alpha, Kinv, ld, ctx->d_prog, partial, n_hyp, xb, nb);
    };
    if (gen)
        go(dmll_kernel<T, 0, RECT>);
    else if (d <= 4)
        go(dmll_kernel<T, 4, RECT>);
    else if (d <= 8)
        go(dmll_kernel<T, 8, RECT>);
    else if (d <= 16)
        go(dmll_kernel<T, 16, RECT>);
    else
        go(dmll_kernel<T, 32, RECT>);
    return (int64_t)ntr * ntc;
}
template <typename T>
int64_t launch_dmll(gpmi_ctx* ctx, const T* x, int64_t n, int d, const T* alpha, const T* Kinv, int64_t ld, double* partial,
                    int n_hyp, bool kinv_negated) {
    return launch_dmll_any<T, false>(ctx, x, n, d, alpha, Kinv, ld, partial, n_hyp, x, kinv_negated ? -1 : n);
}
template <typename T>
int64_t launch_dmll_rect(gpmi_ctx* ctx, const T* xa, int64_t na, const T* xb, int64_t nb, int d, const T* Wt, int64_t ld,
                         double* partial, int n_hyp) {
    return launch_dmll_any<T, true>(ctx, xa, na, d, xa /* unused */, Wt, ld, partial, n_hyp, xb, nb);
}

void launch_reduce_partials(gpmi_ctx* ctx, const double* partial, int64_t nblocks, int nslots, double* out) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)nslots), dim3(1024), 0, ctx->stream, partial, nblocks, nslots, out);
}

template void launch_set_identity<double>(gpmi_ctx*, double*, int64_t, int64_t);
template void launch_set_identity<float>(gpmi_ctx*, float*, int64_t, int64_t);
template int64_t launch_dmll<double>(gpmi_ctx*, const double*, int64_t, int, const double*, const double*, int64_t, double*, int, bool);
template int64_t launch_dmll<float>(gpmi_ctx*, const float*, int64_t, int, const float*, const float*, int64_t, double*, int, bool);
template int64_t launch_dmll_rect<double>(gpmi_ctx*, const double*, int64_t, const double*, int64_t, int, const double*, int64_t, double*, int);
template int64_t launch_dmll_rect<float>(gpmi_ctx*, const float*, int64_t, const float*, int64_t, int, const float*, int64_t, double*, int);

}  // namespace gpmi
