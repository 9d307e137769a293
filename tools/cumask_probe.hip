// CU-masked streams on gfx950: does hipExtStreamCreateWithCUMask work, where do the workgroups land (XCC / SE / CU
// from the hardware id registers), and do two masked streams run concurrently?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>

__global__ void where_kernel(unsigned* out, long long spin) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hwid;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

int main(int argc, char** argv) {
    int want = argc > 1 ? atoi(argv[1]) : 32;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    printf("CUs %d, wall clock rate %d kHz\n", ncu, prop.clockRate);
    int stride = ncu / want;
    std::vector<uint32_t> pm((ncu + 31) / 32, 0u), gm((ncu + 31) / 32, 0u);
    int np = 0;
    for (int i = 0; i < ncu; ++i) {
        bool panel = (i % stride) == ((i / stride) % stride);
        (panel ? pm : gm)[i / 32] |= 1u << (i % 32);
        np += panel;
    }
    hipStream_t sg = nullptr, sp = nullptr, sf = nullptr;
    CK(hipExtStreamCreateWithCUMask(&sg, (uint32_t)gm.size(), gm.data()));
    CK(hipExtStreamCreateWithCUMask(&sp, (uint32_t)pm.size(), pm.data()));
    CK(hipStreamCreateWithFlags(&sf, hipStreamNonBlocking));
    printf("panel CUs %d, streams %p %p\n", np, (void*)sg, (void*)sp);
    unsigned* d;
    const int G = 2048;
    CK(hipMalloc(&d, G * 2 * sizeof(unsigned)));
    std::vector<unsigned> h(G * 2);
    auto report = [&](const char* name, hipStream_t s, int grid) {
        CK(hipMemset(d, 0xff, G * 2 * sizeof(unsigned)));
        hipLaunchKernelGGL(where_kernel, dim3(grid), dim3(256), 0, s, d, 200000LL);  // ~2 ms at 100 MHz
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, G * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::map<unsigned, int> perx;
        std::map<unsigned, int> percu;
        for (int b = 0; b < grid; ++b) {
            unsigned hw = h[2 * b], x = h[2 * b + 1] & 0xf;
            perx[x]++;
            unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            percu[(x << 12) | (se << 8) | (sh << 4) | cu]++;
        }
        printf("%s: grid %d ->", name, grid);
        for (auto& kv : perx) printf(" xcc%u:%d", kv.first, kv.second);
        printf(" | distinct (xcc,se,sh,cu) = %zu\n", percu.size());
        printf("   first blocks xcc:");
        for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b + 1] & 0xf);
        printf("\n");
    };
    report("full ", sf, 512);
    if (sg) report("gemm ", sg, 2 * (ncu - np));
    if (sp) report("panel", sp, 2 * np);
    // concurrency: long kernel on gemm stream, short chain on panel stream
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode) {
        hipStream_t s1 = mode == 0 ? sf : sg, s2 = mode == 0 ? sf : sp;
        if (mode == 2) { s1 = sf; s2 = sp; }
        if (!s1 || !s2) continue;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, s1));
        hipLaunchKernelGGL(where_kernel, dim3(mode == 0 ? 512 : 2 * (ncu - np)), dim3(256), 0, s1, d, 100000LL);  // 1 ms
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(where_kernel, dim3(1), dim3(256), 0, s2, d + 2000, 5000LL);  // 20 x 50 us
        CK(hipStreamSynchronize(s2));
        CK(hipEventRecord(b, s1));
        CK(hipStreamSynchronize(s1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("mode %d (%s): long kernel + 20 short kernels -> %.3f ms (serial would be ~2 ms, overlapped ~1 ms)\n", mode,
               mode == 0 ? "same full stream" : mode == 1 ? "gemm-mask + panel-mask" : "full + panel-mask", ms);
    }
    return 0;
}
