// Memory-pattern probe for the strip-walking producers (LABNOTES.md §12.8): the same strip walk (workgroup = 16 channels x 64 columns,
// 32-row segments, three rows requested ahead, buffers rotating by name) copying a (8, 32, 1024, 1024) fp32 tensor into F-form
// records, once from NCHW planes (256-byte pieces per channel row, 16 planes per workgroup and row: what the blur reads today) and
// once from F-form records (4 KB per workgroup and row: what it would read if the transposed convs wrote a channel-blocked z),
// against a linear float4 copy.  Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pp tools/probes/pattern_probe.hip && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int B = 8, C = 32, H = 1024, W = 1024, KC = C / 16, SEG = 32, NSTRIP = W / 64, NSEG = H / SEG;

__global__ __launch_bounds__(256) void walk_nchw(const float* __restrict__ src, float* __restrict__ dst) {
    __shared__ __attribute__((aligned(16))) float gat[2][16][68];
    int w = blockIdx.x;
    const int strip = w % NSTRIP; w /= NSTRIP;
    const int seg = w % NSEG; w /= NSEG;
    const int kc = w % KC, b = w / KC;
    const int tid = threadIdx.x, ch = tid >> 4, q = tid & 15, X0 = 64 * strip, Y0 = seg * SEG;
    const float* p = src + ((long)(b * C + kc * 16 + ch) * H) * W + X0 + 4 * q;
    float* d = dst + ((long)(b * KC + kc) * H) * W * 16;
    const int pcol = tid >> 2, pq = tid & 3;
    float4 r0 = *reinterpret_cast<const float4*>(p + (long)(Y0 + 0) * W);
    float4 r1 = *reinterpret_cast<const float4*>(p + (long)(Y0 + 1) * W);
    float4 r2 = *reinterpret_cast<const float4*>(p + (long)(Y0 + 2) * W);
    auto step = [&](int Y, float4& r) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        const float4 v = r;
        r = *reinterpret_cast<const float4*>(p + (long)min(Y + 3, H - 1) * W);
        float (*gb)[68] = gat[Y & 1];
        *reinterpret_cast<float4*>(&gb[ch][4 * q]) = v;
        __syncthreads();
        *reinterpret_cast<float4*>(d + ((long)Y * W + X0 + pcol) * 16 + 4 * pq) =
            make_float4(gb[4 * pq][pcol], gb[4 * pq + 1][pcol], gb[4 * pq + 2][pcol], gb[4 * pq + 3][pcol]);
    };
    for (int Y = Y0; Y + 2 < Y0 + SEG; Y += 3) { step(Y, r0); step(Y + 1, r1); step(Y + 2, r2); }
    step(Y0 + SEG - 2, r0); step(Y0 + SEG - 1, r1);      // 32 = 10 * 3 + 2
}

__global__ __launch_bounds__(256) void walk_fform(const float* __restrict__ src, float* __restrict__ dst) {
    int w = blockIdx.x;
    const int strip = w % NSTRIP; w /= NSTRIP;
    const int seg = w % NSEG; w /= NSEG;
    const int kc = w % KC, b = w / KC;
    const int tid = threadIdx.x, X0 = 64 * strip, Y0 = seg * SEG;
    const long base = ((long)(b * KC + kc) * H) * W * 16 + (long)(X0 + (tid >> 2)) * 16 + 4 * (tid & 3);
    float4 r0 = *reinterpret_cast<const float4*>(src + base + (long)(Y0 + 0) * W * 16);
    float4 r1 = *reinterpret_cast<const float4*>(src + base + (long)(Y0 + 1) * W * 16);
    float4 r2 = *reinterpret_cast<const float4*>(src + base + (long)(Y0 + 2) * W * 16);
    auto step = [&](int Y, float4& r) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        const float4 v = r;
        r = *reinterpret_cast<const float4*>(src + base + (long)min(Y + 3, H - 1) * W * 16);
        *reinterpret_cast<float4*>(dst + base + (long)Y * W * 16) = v;
    };
    for (int Y = Y0; Y + 2 < Y0 + SEG; Y += 3) { step(Y, r0); step(Y + 1, r1); step(Y + 2, r2); }
    step(Y0 + SEG - 2, r0); step(Y0 + SEG - 1, r1);
}

__global__ __launch_bounds__(256) void linear_copy(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}

int main() {
    const long n = (long)B * C * H * W;
    float *a, *f, *d;
    hipMalloc(&a, n * 4); hipMalloc(&f, n * 4); hipMalloc(&d, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(f, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = B * KC * NSTRIP * NSEG;
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-34s %8.1f us  %7.1f GB/s (read + write %.2f GB)\n", name, ms * 1e3, 2.0 * n * 4 / ms / 1e6, 2.0 * n * 4 / 1e9);
    };
    time("strip walk, NCHW planes in", [&] { hipLaunchKernelGGL(walk_nchw, dim3(grid), dim3(256), 0, 0, a, d); });
    time("strip walk, F-form records in", [&] { hipLaunchKernelGGL(walk_fform, dim3(grid), dim3(256), 0, 0, f, d); });
    time("linear float4 copy", [&] { hipLaunchKernelGGL(linear_copy, dim3(256 * 32), dim3(256), 0, 0, (const float4*)a, (float4*)d, n / 4); });
    time("strip walk, NCHW planes in", [&] { hipLaunchKernelGGL(walk_nchw, dim3(grid), dim3(256), 0, 0, a, d); });
    time("strip walk, F-form records in", [&] { hipLaunchKernelGGL(walk_fform, dim3(grid), dim3(256), 0, 0, f, d); });
    return 0;
}
