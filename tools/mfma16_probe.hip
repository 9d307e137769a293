// accumulator layout of v_mfma_f64_16x16x4_f64: which (row, col) does register r of lane l hold?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int l = threadIdx.x, i = l & 15, k = l >> 4;
    // pass 1: A[i][k] = (i + 1) for k == 0 else 0 ; B[j][k] = 1 for k == 0  ->  D[i][j] = i + 1      (row of each slot)
    // pass 2: A[i][k] = 1 for k == 0 ; B[j][k] = (j + 1) for k == 0          ->  D[i][j] = j + 1      (col of each slot)
    // pass 3: A[i][k] = (k + 1), B[j][k] = 1                                ->  D = 1 + 2 + 3 + 4 = 10 everywhere (k mapping sane)
    v4d z = {0, 0, 0, 0};
    v4d d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(k == 0 ? (double)(i + 1) : 0.0, k == 0 ? 1.0 : 0.0, z, 0, 0, 0);
    v4d d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(k == 0 ? 1.0 : 0.0, k == 0 ? (double)(i + 1) : 0.0, z, 0, 0, 0);
    v4d d3 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(k + 1), 1.0, z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        out[(l * 4 + r) * 3 + 0] = d1[r];
        out[(l * 4 + r) * 3 + 1] = d2[r];
        out[(l * 4 + r) * 3 + 2] = d3[r];
    }
}
int main() {
    double* d; hipMalloc(&d, 64 * 4 * 3 * sizeof(double));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[64 * 4 * 3]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            int row = (int)h[(l * 4 + r) * 3] - 1, col = (int)h[(l * 4 + r) * 3 + 1] - 1;
            if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> row %2d col %2d  (sum-k check %.0f)\n", l, r, row, col, h[(l * 4 + r) * 3 + 2]);
            if (row != 4 * (l >> 4) + r || col != (l & 15)) ok = 0;
        }
    printf("layout row = 4*(lane>>4)+reg, col = lane&15 : %s\n", ok ? "YES" : "NO");
    return 0;
}
