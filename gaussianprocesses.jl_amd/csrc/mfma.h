// mfma.h — per-precision matrix-core traits shared by the trailing-update kernel (gemm.hip) and the
// panel kernels (panel.hip).  One 16 x 16 x 4 product per instruction in both precisions:
// v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 (the accumulator layouts differ, see row_of).
//   operand layout (both):  lane l supplies A[row = l & 15][k = l >> 4],  B[col = l & 15][k = l >> 4]
//   result layout:          element r of lane l is C[row_of(l, r)][col_of(l, r)]
#pragma once
#include <hip/hip_runtime.h>

namespace gpmi {

template <typename T>
struct Mfma;
template <int CTRL>
__device__ __forceinline__ double dpp_rot(double x) {
    // rotate within each 16-lane row; row_ror:n gives out[l] = in[(l - n) & 15]  (probed, tools/dpp_probe.hip)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// The four-instruction form (kept for the ablation, VARIANT & 512 of the update kernel): it was the first choice because a
// bare-loop micro-benchmark put v_mfma_f64_16x16x4 at 48 TFLOP/s against 69 for v_mfma_f64_4x4x4_4b; inside the real
// kernel the one-instruction form is 5 % (two workgroups per CU) to 11 % (one) FASTER — no B rotations, a quarter of the
// MFMA issue slots — and it is what rocBLAS's MT128x128x16_MI16x16x4 kernel uses.
struct Mfma4x4d {
    using Vec = double __attribute__((ext_vector_type(2)));
    static constexpr int E = 2;
    static constexpr int BK = 16;
    static constexpr int NR = 4;  // accumulator registers per 16 x 16 tile
    static constexpr bool NEG = false;
    struct Acc {
        double v[4];
    };
    // one 16 x 16 x 4 product = four 4x4x4 (4-block) MFMAs against B rotated by r blocks
    static __device__ __forceinline__ void rotations(double b, double (&br)[4]) {
        br[0] = b;
        br[1] = dpp_rot<0x12C>(b);  // row_ror:12 -> lane l reads l + 4  (block b + 1)
        br[2] = dpp_rot<0x128>(b);  // row_ror:8                        (block b + 2)
        br[3] = dpp_rot<0x124>(b);  // row_ror:4  -> lane l reads l + 12 (block b + 3)
    }
    static __device__ __forceinline__ void mma(double a, const double (&br)[4], Acc& c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c.v[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, br[r], c.v[r], 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_sub(double a, const double (&br)[4], Acc& c) { mma(a, br, c); }
    // acc register r of lane (j + 4b + 16i) holds C[row 4b + i][col 4((b + r) & 3) + j]
    static __device__ __forceinline__ int row_of(int lane, int) { return ((lane >> 2) & 3) * 4 + (lane >> 4); }
    static __device__ __forceinline__ int col_of(int lane, int r) { return ((((lane >> 2) & 3) + r) & 3) * 4 + (lane & 3); }
};
// fp64: one v_mfma_f64_16x16x4_f64 per 16 x 16 x 4 product.
template <>
struct Mfma<double> {
    using Vec = double __attribute__((ext_vector_type(2)));
    static constexpr int E = 2;
    static constexpr int BK = 16;
    static constexpr int NR = 4;
    using V4 = double __attribute__((ext_vector_type(4)));
    struct Acc {
        V4 v;
    };
    static __device__ __forceinline__ void rotations(double b, double (&br)[4]) { br[0] = b; }
    static __device__ __forceinline__ void mma(double a, const double (&br)[4], Acc& c) {
        c.v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, br[0], c.v, 0, 0, 0);
    }
    // c -= a b': the fp64 MFMAs reuse the BLGP field as neg:[a, b, c] modifiers, so the subtraction is free
    static constexpr bool NEG = true;
    static __device__ __forceinline__ void mma_sub(double a, const double (&br)[4], Acc& c) {
        c.v = __builtin_amdgcn_mfma_f64_16x16x4f64(a, br[0], c.v, 0, 0, 1);
    }
    // probed on hardware (tools/mfma16_probe.hip): register r of lane l holds D[row 4 r + (l >> 4)][col l & 15]
    // (NOT the fp32 instruction's 4 (l >> 4) + r)
    static constexpr int RSTEP = 4;  // rows between consecutive accumulator registers
    static __device__ __forceinline__ int row_of(int lane, int reg) { return 4 * reg + (lane >> 4); }
    static __device__ __forceinline__ int col_of(int lane, int) { return lane & 15; }
};

template <>
struct Mfma<float> {
    using Vec = float __attribute__((ext_vector_type(4)));
    static constexpr int E = 4;
    static constexpr int BK = 32;
    static constexpr int NR = 4;
    using V4 = float __attribute__((ext_vector_type(4)));
    struct Acc {
        V4 v;
    };
    static __device__ __forceinline__ void rotations(float b, float (&br)[4]) { br[0] = b; }
    static __device__ __forceinline__ void mma(float a, const float (&br)[4], Acc& c) {
        c.v = __builtin_amdgcn_mfma_f32_16x16x4f32(a, br[0], c.v, 0, 0, 0);
    }
    // (no neg modifiers on the fp32 MFMAs: callers that want c - a b' keep -c in the accumulator, see gemm.hip)
    static constexpr bool NEG = false;
    static __device__ __forceinline__ void mma_sub(float a, const float (&br)[4], Acc& c) { mma(a, br, c); }
    // v_mfma_f32_16x16x4_f32 C/D map: col = lane & 15, row = 4 * (lane >> 4) + reg
    static constexpr int RSTEP = 1;
    static __device__ __forceinline__ int row_of(int lane, int reg) { return 4 * (lane >> 4) + reg; }
    static __device__ __forceinline__ int col_of(int lane, int) { return lane & 15; }
};
template <typename T>
__device__ __forceinline__ T acc_get(const typename Mfma<T>::Acc& a, int r);
template <>
__device__ __forceinline__ double acc_get<double>(const Mfma<double>::Acc& a, int r) { return a.v[r]; }
template <>
__device__ __forceinline__ float acc_get<float>(const Mfma<float>::Acc& a, int r) { return a.v[r]; }


// acc += A[16 x K] * B[16 x K]'  for operands held row-major (k contiguous) in LDS or global memory
template <typename T>
__device__ __forceinline__ void mma16_nt(typename Mfma<T>::Acc& acc, const T* A, int lda, const T* B, int ldb, int K, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    const T* ap = A + r16 * lda + g;
    const T* bp = B + r16 * ldb + g;
#pragma unroll 4
    for (int kk = 0; kk < K; kk += 4) {
        T br[4];
        Mfma<T>::rotations(bp[kk], br);
        Mfma<T>::mma(ap[kk], br, acc);
    }
}
template <typename T>
__device__ __forceinline__ void acc_zero(typename Mfma<T>::Acc& acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc.v[r] = T(0);
}

}  // namespace gpmi
