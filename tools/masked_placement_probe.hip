// Where do the workgroups of a persistent launch land, and when do they start, on the library's CU-masked update stream (one CU per
// XCD reserved: mask bit 33 k) against a plain stream?  Two shapes: the 256 x 128 update (512 threads, 144 KiB of LDS: one
// workgroup per CU, grid 248) and the 128 x 128 one (256 threads, 64 KiB, 256 VGPRs: two per CU, grid 496).  Each workgroup records
// its XCC / SE / CU (hardware id registers) and its start time, then holds its CU for ~1 ms.
//   hipcc --offload-arch=gfx950 -O2 tools/masked_placement_probe.hip -o tools/bin/masked_placement_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

struct Rec {
    unsigned hwid, xcc;
    long long t0;
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void hold_kernel(Rec* out, long long spin) {
    extern __shared__ unsigned char lds[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    lds[threadIdx.x] = (unsigned char)threadIdx.x;  // the LDS allocation is real
    __syncthreads();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) {
        out[blockIdx.x].hwid = hwid;
        out[blockIdx.x].xcc = xcc + (lds[5] == 77 ? 1000u : 0u);
        out[blockIdx.x].t0 = t0;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int THREADS>
static int run(const char* name, hipStream_t s, int grid, unsigned ldsbytes, Rec* d, std::vector<Rec>& h) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hold_kernel<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsbytes));
    CK(hipMemset(d, 0, h.size() * sizeof(Rec)));
    hipLaunchKernelGGL(hold_kernel<THREADS>, dim3(grid), dim3(THREADS), ldsbytes, s, d, 100000LL);  // 1 ms at 100 MHz
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
    long long tmin = h[0].t0;
    for (int b = 0; b < grid; ++b) tmin = std::min(tmin, h[b].t0);
    std::map<unsigned, int> perx, percu;
    int match = 0, late = 0;
    for (int b = 0; b < grid; ++b) {
        const unsigned x = h[b].xcc & 0xf, hw = h[b].hwid;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        perx[x]++;
        percu[(x << 12) | (se << 8) | (sh << 4) | cu]++;
        match += (int)(x == (unsigned)(b & 7));
        late += (h[b].t0 - tmin) > 50000;  // started more than 0.5 ms after the first: it waited for another workgroup's CU
    }
    int maxper = 0;
    for (auto& kv : percu) maxper = std::max(maxper, kv.second);
    printf("%-34s grid %3d: per XCC", name, grid);
    for (auto& kv : perx) printf(" %d", kv.second);
    printf(" | blockIdx %% 8 == XCC for %d | distinct CUs %zu, at most %d workgroups on one | started late %d\n", match, percu.size(), maxper, late);
    if (getenv("PROBE_DETAIL")) {
        // per XCC: workgroups per shader engine, and the CUs (se.sh.cu) that got more than one workgroup
        for (unsigned x = 0; x < 8; ++x) {
            int perse[8] = {0};
            printf("   xcc %u: per SE", x);
            for (auto& kv : percu)
                if ((kv.first >> 12) == x) perse[(kv.first >> 8) & 7] += kv.second;
            for (int e = 0; e < 8; ++e)
                if (perse[e]) printf(" se%d:%d", e, perse[e]);
            printf(" | CUs per SE");
            int cus[8] = {0};
            for (auto& kv : percu)
                if ((kv.first >> 12) == x) cus[(kv.first >> 8) & 7]++;
            for (int e = 0; e < 8; ++e)
                if (cus[e]) printf(" se%d:%d", e, cus[e]);
            printf(" | doubled:");
            for (auto& kv : percu)
                if ((kv.first >> 12) == x && kv.second > 1) printf(" %u.%u.%u", (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15);
            printf("\n");
        }
        printf("   XCC of blocks 0..15:");
        for (int b = 0; b < 16; ++b) printf(" %u", h[b].xcc & 0xf);
        printf("\n");
    }
    return 0;
}

int main() {
    uint32_t side_m[8] = {0}, upd_m[8];
    for (int k = 0; k < 8; ++k) side_m[(k * 33) / 32] |= 1u << ((k * 33) % 32);
    for (int w = 0; w < 8; ++w) upd_m[w] = ~side_m[w];
    hipStream_t plain = nullptr, masked = nullptr;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    CK(hipExtStreamCreateWithCUMask(&masked, 8, upd_m));
    Rec* d = nullptr;
    std::vector<Rec> h(1024);
    CK(hipMalloc(&d, h.size() * sizeof(Rec)));
    for (int rep = 0; rep < 1; ++rep) {
        if (run<512>("plain  512 thr 144 KiB (1 per CU)", plain, 248, 147456, d, h)) return 1;
        if (run<512>("masked 512 thr 144 KiB (1 per CU)", masked, 248, 147456, d, h)) return 1;
        if (run<256>("plain  256 thr  64 KiB (2 per CU)", plain, 496, 65536, d, h)) return 1;
        if (run<256>("masked 256 thr  64 KiB (2 per CU)", masked, 496, 65536, d, h)) return 1;
        if (run<512>("masked 512 thr 144 KiB, grid 240", masked, 240, 147456, d, h)) return 1;
    }
    return 0;
}
