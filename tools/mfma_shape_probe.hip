// Probe: the tap body of the 8-wave split-f16 conv kernels with the two f16 MFMA shapes, under the kernel's own conditions —
// 8 waves per workgroup (two per SIMD), one workgroup per CU, every operand fragment re-read from LDS by ds_read_b128, the
// reads of tap i+1 issued before the MFMAs of tap i (two register sets), random data.
//   X: v_mfma_f32_32x32x16_f16, wave tile 64 channels x 2 rows x 32 pixels: 8 fragment reads + 12 MFMAs per tap
//   Y: v_mfma_f32_16x16x32_f16 with the hi/lo halves as the K = 32 halves ([w_hi|w_hi] x [x_hi;x_lo], and the lo x hi products
//      of two taps paired): 12 fragment reads + 24 MFMAs per tap, the same products and the same accumulator registers
// Both do 12 x 16384 = 24 x 8192 MACs per wave and tap.  hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)

constexpr int LDS_BYTES = 150 * 1024;

template <int SHAPE>
__global__ __launch_bounds__(512) void tap_loop(const uint4* __restrict__ src, float* __restrict__ out, int ntaps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = src[i];
    __syncthreads();
    // per-lane fragment base: conflict-free 16-byte stride, per-wave row offset as in the conv kernels
    const unsigned char* lw = smem + 80 * 1024 + lane * 16;
    const unsigned char* lx = smem + wave * 2 * 2176 + lane * 16;
    if (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        struct F { half8 ah[2], al[2], bh[2], bl[2]; };
        auto load = [&](F& f, int t) {
            const int o = (t % 9) * 4096;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f.ah[m] = *reinterpret_cast<const half8*>(lw + o + m * 1024);
                f.al[m] = *reinterpret_cast<const half8*>(lw + o + 2048 + m * 1024);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f.bh[n] = *reinterpret_cast<const half8*>(lx + (t % 3) * 64 + n * 2176);
                f.bl[n] = *reinterpret_cast<const half8*>(lx + (t % 3) * 64 + n * 2176 + 32 * 1024);
            }
        };
        auto mfma = [&](const F& f) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[m], f.bh[n], acc[m][n], 0, 0, 0);
        };
        F f0, f1;
        load(f0, 0);
        SB();
        for (int t = 0; t < ntaps; t += 2) {
            load(f1, t + 1); SB(); mfma(f0); SB();
            load(f0, t + 2); SB(); mfma(f1); SB();
        }
        float s = 0.f;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int r = 0; r < 16; ++r) s += acc[a][b][r];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
                for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
        // per tap: A1[4] = [w_hi|w_hi] of the 4 M-tiles, B1[4] = [x_hi;x_lo] of the 4 N-tiles; per tap PAIR: A2[4] = [w_lo(a)|w_lo(b)],
        // B2[4] = [x_hi(a);x_hi(b)] — loaded half per tap (2 + 2)
        struct F { half8 a1[4], b1[4], a2[2], b2[2]; };
        auto load = [&](F& f, int t) {
            const int o = (t % 9) * 4096;
#pragma unroll
            for (int m = 0; m < 4; ++m) f.a1[m] = *reinterpret_cast<const half8*>(lw + o + m * 1024);
#pragma unroll
            for (int n = 0; n < 4; ++n) f.b1[n] = *reinterpret_cast<const half8*>(lx + (t % 3) * 64 + (n >> 1) * 2176 + (n & 1) * 1024);
#pragma unroll
            for (int m = 0; m < 2; ++m) f.a2[m] = *reinterpret_cast<const half8*>(lw + o + 40 * 1024 + m * 1024);
#pragma unroll
            for (int n = 0; n < 2; ++n) f.b2[n] = *reinterpret_cast<const half8*>(lx + (t % 3) * 64 + 32 * 1024 + n * 1024);
        };
        auto mfma = [&](const F& f, const F& g) {       // g: the partner tap's half of the paired fragments
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a1[m], f.b1[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a2[m], f.b2[n], acc[m][n], 0, 0, 0);
                    acc[m + 2][n + 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g.a2[m], g.b2[n], acc[m + 2][n + 2], 0, 0, 0);
                }
        };
        F f0, f1;
        load(f0, 0);
        load(f1, 1);
        SB();
        for (int t = 0; t < ntaps; t += 2) {
            // 24 MFMAs per tap: 16 + 8 (half of the pair's 16 lo x hi products per tap)
            SB(); mfma(f0, f1); SB();
            load(f0, t + 2); SB(); mfma(f1, f0); SB();
            load(f1, t + 3);
        }
        float s = 0.f;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
                for (int r = 0; r < 4; ++r) s += acc[a][b][r];
        out[blockIdx.x * 512 + tid] = s;
    }
}

int main() {
    std::vector<_Float16> h(LDS_BYTES / 2);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) v = (_Float16)(nd(g) * 0.05f);
    uint4* src;
    float* out;
    (void)hipMalloc(&src, LDS_BYTES);
    (void)hipMalloc(&out, 4096 * 512 * 4);
    (void)hipMemcpy(src, h.data(), LDS_BYTES, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tap_loop<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tap_loop<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int ntaps = 9 * 32 * 8, grid = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            (void)hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(tap_loop<0>, dim3(grid), dim3(512), LDS_BYTES, 0, src, out, ntaps);
            else hipLaunchKernelGGL(tap_loop<1>, dim3(grid), dim3(512), LDS_BYTES, 0, src, out, ntaps);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double macs = (double)grid * 8 * ntaps * 12 * 16384.0;
            printf("shape %s: %.3f ms, %.1f TF/s at the MFMA level (3 per useful product), %.1f ns per tap round of a CU's 8 waves\n",
                   shape ? "16x16x32 (12 reads + 24 MFMAs / tap)" : "32x32x16 ( 8 reads + 12 MFMAs / tap)", ms, 2 * macs / ms / 1e9,
                   ms * 1e6 / ((double)grid / 256 * ntaps));
        }
    return 0;
}
