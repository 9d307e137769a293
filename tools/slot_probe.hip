// Which of the two co-resident persistent workgroups am I?  Candidates: HW_ID.wave_id (SIMD wave slot) and LDS_ALLOC.lds_base.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, long long spin) {
    __shared__ double big[8192];  // 64 KiB like the GEMM
    unsigned hwid, xcc, lds;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds));
    big[threadIdx.x] = (double)hwid;
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) {
        unsigned* o = out + 4 * (blockIdx.x * 4 + (threadIdx.x >> 6));
        o[0] = hwid; o[1] = xcc; o[2] = lds; o[3] = (unsigned)big[threadIdx.x];
    }
}
int main() {
    const int G = 512;
    unsigned* d;
    hipMalloc(&d, G * 16 * sizeof(unsigned));
    hipMemset(d, 0, G * 16 * sizeof(unsigned));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, d, 100000LL);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(G * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> waveid, ldsbase;
    std::map<unsigned long long, std::vector<int>> percu;
    for (int b = 0; b < G; ++b)
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[4 * (b * 4 + w)], x = h[4 * (b * 4 + w) + 1] & 0xf, l = h[4 * (b * 4 + w) + 2];
            waveid[hw & 0xf]++;
            ldsbase[l & 0xfff]++;
            if (w == 0) percu[((unsigned long long)x << 32) | (hw & 0xff00)].push_back(b);
        }
    printf("wave_id histogram:");
    for (auto& kv : waveid) printf(" %u:%d", kv.first, kv.second);
    printf("\nlds_alloc[11:0] histogram:");
    for (auto& kv : ldsbase) printf(" 0x%x:%d", kv.first, kv.second);
    printf("\nCUs seen %zu; first CUs -> blocks:", percu.size());
    int k = 0;
    for (auto& kv : percu) {
        if (k++ >= 12) break;
        printf(" [");
        for (int b : kv.second) printf("%d ", b);
        printf("]");
    }
    printf("\nsample raw: hwid %08x lds %08x | hwid %08x lds %08x\n", h[0], h[2], h[16 * 256], h[16 * 256 + 2]);
    return 0;
}
