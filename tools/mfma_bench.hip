// mfma_bench.hip — what is the real FP64 matrix/vector ceiling of this MI355X?
// Measures cycles per instruction (s_memtime) and chip throughput for
//   v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64, v_fma_f64 (VALU), and MFMA+VALU mixed,
// at 1/2/4 waves per SIMD with 1..8 independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(double* out, long long* cyc, int iters) {
    d4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.0;
    double x = 1.0 + (threadIdx.x & 7) * 0.125, y = 0.5 + (threadIdx.x & 3) * 1e-3;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    if (s == -1.2345) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma4(double* out, long long* cyc, int iters) {
    double acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = 0.0;
    double x = 1.0 + (threadIdx.x & 7) * 0.125, y = 0.5 + (threadIdx.x & 3) * 1e-3;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0; for (int a = 0; a < NACC; ++a) s += acc[a];
    if (s == -1.2345) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, long long* cyc, int iters) {
    double acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = threadIdx.x * 1e-3 + a;
    double x = 1.0 + 1e-9 * (threadIdx.x & 7), y = 1e-7 * (threadIdx.x & 3);
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = fma(acc[a], x, y);
    }
    long long t1 = clock64();
    double s = 0; for (int a = 0; a < NACC; ++a) s += acc[a];
    if (s == -1.2345) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// mixed: waves 0..1 of each block do MFMA, waves 2..3 do VALU fma
__global__ __launch_bounds__(256) void k_mixed(double* out, long long* cyc, int iters) {
    double s = 0;
    if ((threadIdx.x >> 6) < 2) {
        d4 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.0;
        double x = 1.0 + (threadIdx.x & 7) * 0.125, y = 0.5;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    } else {
        double acc[8];
        for (int a = 0; a < 8; ++a) acc[a] = threadIdx.x * 1e-3 + a;
        double x = 1.0 + 1e-9, y = 1e-7;
        for (int i = 0; i < iters * 8; ++i)
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = fma(acc[a], x, y);
        for (int a = 0; a < 8; ++a) s += acc[a];
    }
    if (s == -1.2345) out[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F>
int run(const char* name, F kern, int blocks, int iters, double flops_per_wave_iter, int ninstr_per_iter) {
    double* out; long long* cyc; CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    double tf = (double)blocks * 4 * iters * flops_per_wave_iter / (ms * 1e-3) / 1e12;
    printf("%-34s blocks=%5d  %8.3f ms  %7.2f TFLOP/s   wave0: %6.1f cyc/instr  (s_memtime clk ~ %.0f MHz)\n", name, blocks, ms, tf,
           (double)h / ((double)iters * ninstr_per_iter), (double)h / (ms * 1e-3) / 1e6);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    const int it = 20000;
    const double F16 = 2.0 * 16 * 16 * 4, F4 = 2.0 * 4 * 4 * 4 * 4, FV = 2.0 * 64;
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        printf("--- %d wave(s) per SIMD\n", wps);
        run("mfma_f64_16x16x4  1 acc", k_mfma16<1>, blocks, it, F16 * 1, 1);
        run("mfma_f64_16x16x4  2 acc", k_mfma16<2>, blocks, it, F16 * 2, 2);
        run("mfma_f64_16x16x4  4 acc", k_mfma16<4>, blocks, it, F16 * 4, 4);
        run("mfma_f64_16x16x4  8 acc", k_mfma16<8>, blocks, it, F16 * 8, 8);
        run("mfma_f64_4x4x4_4b 4 acc", k_mfma4<4>, blocks, it, F4 * 4, 4);
        run("mfma_f64_4x4x4_4b 8 acc", k_mfma4<8>, blocks, it, F4 * 8, 8);
        run("v_fma_f64         4 acc", k_fma<4>, blocks, it * 4, FV * 4, 4);
        run("v_fma_f64         8 acc", k_fma<8>, blocks, it * 4, FV * 8, 8);
    }
    // a single CU's worth of work (no chip-level power pressure): 1 block
    printf("--- one workgroup only (1 CU busy)\n");
    run("mfma_f64_16x16x4  4 acc", k_mfma16<4>, 1, it, F16 * 4, 4);
    run("v_fma_f64         8 acc", k_fma<8>, 1, it * 4, FV * 8, 8);
    {
        double* out; long long* cyc; hipMalloc(&out, 64); hipMalloc(&cyc, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_mixed, dim3(512), dim3(256), 0, 0, out, cyc, 64); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mixed, dim3(512), dim3(256), 0, 0, out, cyc, it);
        hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        double fm = 512.0 * 2 * it * 4 * F16, fv = 512.0 * 2 * it * 8.0 * 8 * FV;
        printf("mixed (2 MFMA waves + 2 VALU waves per WG, 2 WG/CU): %.3f ms  mfma %.2f TF + valu %.2f TF = %.2f TF\n", ms,
               fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
    }
    return 0;
}
