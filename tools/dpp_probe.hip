#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x124, 0xF, 0xF, false);        // row_ror:4
    out[64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x12C, 0xF, 0xF, false);   // row_ror:12
    out[128 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0x128, 0xF, 0xF, false);  // row_ror:8
}
int main() {
    int* d; hipMalloc(&d, 192 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int t = 0; t < 3; ++t) { printf("%s:", t == 0 ? "ror4 " : t == 1 ? "ror12" : "ror8 "); for (int l = 0; l < 20; ++l) printf(" %d", h[t * 64 + l]); printf(" ... lane48: %d\n", h[t*64+48]); }
    return 0;
}
