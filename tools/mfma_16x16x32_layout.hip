// Lane layout of v_mfma_f32_16x16x32_f16 as conv_f16s_big.hip assumes it (A: lane l = row l%16, K group l/16 of 8 values; B: column l%16,
// K group l/16; D: column l%16, rows 4*(l/16)+r), checked against a host product.  hipcc --offload-arch=gfx950 -O2 tools/mfma_16x16x32_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// D[16x16] = A[16x32] * B[32x16]; assumed layout: A lane l: row l%16, k = 8*(l/16)+j; B lane l: col l%16, k = 8*(l/16)+j; D: col l%16, rows 4*(l/16)+r
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[i * 32 + 8 * g + j]; b[j] = (_Float16)B[(8 * g + j) * 16 + i]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}
int main() {
    std::vector<float> A(512), B(512), D(256);
    for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7 + i / 32) % 9 - 4); B[i] = (float)((i * 5 + i / 16) % 7 - 3); }
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dD, 1024);
    (void)hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        float ref = 0; for (int kk = 0; kk < 32; ++kk) ref += A[m * 32 + kk] * B[kk * 16 + n];
        if (ref != D[m * 16 + n]) ++bad;
    }
    printf("layout check 16x16x32: %d mismatches of 256\n", bad);
    return 0;
}
