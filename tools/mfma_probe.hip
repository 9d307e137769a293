// mfma_probe.hip — empirically determine the lane layout of v_mfma_f64_4x4x4_4b_f64 (with and without
// the A-broadcast controls cbsz/abid) and its throughput with broadcast, on this MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int CBSZ, int ABID>
__global__ void probe(unsigned long long* out) {
    const int lane = threadIdx.x;
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63;
    double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    unsigned long long m = __ballot(d != 0.0);
    if (lane == 0) out[blockIdx.x] = m;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_bcast(double* out, int iters) {
    double acc[NACC][4];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.0;
    double x = 1.0 + (threadIdx.x & 7) * 0.125, y = 0.5 + (threadIdx.x & 3) * 1e-3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[a][0], 2, 0, 0);
            acc[a][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[a][1], 2, 1, 0);
            acc[a][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[a][2], 2, 2, 0);
            acc[a][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[a][3], 2, 3, 0);
        }
    }
    double s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    if (s == -1.2345) out[0] = s;
}

template <int CBSZ, int ABID>
int run_probe(const char* name) {
    unsigned long long* d; hipMalloc(&d, 4096 * 8);
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(4096), dim3(64), 0, 0, d);
    std::vector<unsigned long long> h(4096);
    hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    int bad = 0;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        int i = la & 3, ba = (la >> 2) & 3, ka = la >> 4, j = lb & 3, bb = (lb >> 2) & 3, kb = lb >> 4;
        unsigned long long expect = 0;
        if (ka == kb) {
            if (CBSZ == 0) { if (ba == bb) expect = 1ull << (j + 4 * bb + 16 * i); }
            else { if (ba == ABID) expect = 1ull << (j + 4 * bb + 16 * i); }
        }
        if (h[la * 64 + lb] != expect) {
            if (bad < 12) printf("  %s: la=%2d lb=%2d got mask %016llx expect %016llx\n", name, la, lb, h[la * 64 + lb], expect);
            ++bad;
        }
    }
    printf("%s: %s (%d mismatches of 4096)\n", name, bad ? "HYPOTHESIS WRONG" : "layout hypothesis CONFIRMED", bad);
    if (bad) {  // dump enough to deduce the real layout
        for (int la = 0; la < 64; la += 1) { printf("   la=%2d:", la); for (int lb = 0; lb < 64; ++lb) if (h[la*64+lb]) printf(" lb%d->%d", lb, __builtin_ctzll(h[la*64+lb])); printf("\n"); if (la > 20) break; }
    }
    return bad;
}

int main() {
    run_probe<0, 0>("cbsz=0");
    run_probe<2, 0>("cbsz=2 abid=0");
    run_probe<2, 1>("cbsz=2 abid=1");
    run_probe<2, 2>("cbsz=2 abid=2");
    run_probe<2, 3>("cbsz=2 abid=3");
    double* out; hipMalloc(&out, 64);
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int nacc = 1; nacc <= 4; nacc *= 2) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int it = 20000; int blocks = 256 * wps;
            auto launch = [&](int iters) {
                if (nacc == 1) hipLaunchKernelGGL(k_bcast<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
                else if (nacc == 2) hipLaunchKernelGGL(k_bcast<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k_bcast<4>, dim3(blocks), dim3(256), 0, 0, out, iters);
            };
            launch(64); hipDeviceSynchronize();
            hipEventRecord(e0); launch(it); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double tf = (double)blocks * 4 * it * nacc * 4 * 512.0 / (ms * 1e-3) / 1e12;
            printf("4x4x4 cbsz=2 bcast (16x16x4-equivalent x%d) waves/SIMD=%d: %.3f ms %.2f TFLOP/s\n", nacc, wps, ms, tf);
        }
    }
    return 0;
}
