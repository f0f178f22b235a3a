// HBM yardstick for the copy-bound kernels of the W+ step (LABNOTES.md §13): what a plain read+write stream of the 1024² activation
// tensor — (8,32,1024,1024) fp32 = 1.07 GB in, 1.07 GB out — reaches on this part, as a function of the number of independent 16-byte
// loads a lane keeps in flight, the store policy and the grid size.  Stand-alone (no torch):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/copy_probe.hip -o tools/probes/build/copy_probe && tools/probes/build/copy_probe
// Prints TB/s counting read + write bytes (the convention of MI355X_MICROARCH.md's 6.29 TB/s float4 copy).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

// U independent 16-byte loads per lane and trip, block-contiguous: a wave's U loads are U consecutive 1 KiB pieces
template <int U, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copy_kernel(const float4v* __restrict__ src, float4v* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * 256 * U;
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
        float4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256;
            const long ic = i < n4 ? i : n4 - 1;
            v[u] = NT_LD ? __builtin_nontemporal_load(src + ic) : src[ic];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256;
            if (i < n4) {
                if (NT_ST) __builtin_nontemporal_store(v[u], dst + i);
                else dst[i] = v[u];
            }
        }
    }
}

// the same with the NEXT trip's loads requested before this trip's stores (two register sets)
template <int U, bool NT_ST>
__global__ __launch_bounds__(256) void copy_pipelined_kernel(const float4v* __restrict__ src, float4v* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * 256 * U;
    long base = (long)blockIdx.x * 256 * U + threadIdx.x;
    float4v a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = src[std::min(base + u * 256, n4 - 1)];
    for (; base < n4; base += stride) {
        const long nb = base + stride;
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = src[std::min(nb + u * 256, n4 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256;
            if (i < n4) {
                if (NT_ST) __builtin_nontemporal_store(a[u], dst + i);
                else dst[i] = a[u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = b[u];
    }
}

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const float4v* __restrict__ src, float* __restrict__ sink, long n4) {
    const long stride = (long)gridDim.x * 256 * U;
    float4v acc = {0, 0, 0, 0};
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
        float4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[std::min(base + u * 256, n4 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int U, bool NT_ST>
__global__ __launch_bounds__(256) void fill_kernel(float4v* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * 256 * U;
    const float4v v = {1.f, 2.f, 3.f, 4.f};
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256;
            if (i < n4) {
                if (NT_ST) __builtin_nontemporal_store(v, dst + i);
                else dst[i] = v;
            }
        }
    }
}

template <typename F>
static double time_us(F launch, int reps = 15) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> t;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const long n = 8L * 32 * 1024 * 1024, n4 = n / 4;
    float4v *src, *dst;
    float* sink;
    CK(hipMalloc(&src, n * 4));
    CK(hipMalloc(&dst, n * 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 0x11, n * 4));
    CK(hipMemset(dst, 0, n * 4));
    const double gb2 = 2.0 * n * 4 / 1e9, gb1 = n * 4 / 1e9;
    printf("tensor (8,32,1024,1024) fp32 = %.3f GB; copy rows count read + write bytes\n", gb1);
    const int grids[] = {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 64};
    auto row = [&](const char* name, double gb, auto launch) {
        printf("%-44s", name);
        for (int g : grids) {
            const double us = time_us([&] { launch(g); });
            printf("  g=%5d %7.1f us %5.2f TB/s", g, us, gb / us * 1e3);
        }
        printf("\n");
        fflush(stdout);
    };
    row("copy U=1 (round 3's linear_copy)", gb2, [&](int g) { hipLaunchKernelGGL((copy_kernel<1, false, false>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=2", gb2, [&](int g) { hipLaunchKernelGGL((copy_kernel<2, false, false>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=4", gb2, [&](int g) { hipLaunchKernelGGL((copy_kernel<4, false, false>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=8", gb2, [&](int g) { hipLaunchKernelGGL((copy_kernel<8, false, false>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=4 nt stores", gb2, [&](int g) { hipLaunchKernelGGL((copy_kernel<4, false, true>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=8 nt stores", gb2, [&](int g) { hipLaunchKernelGGL((copy_kernel<8, false, true>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=4 nt loads + nt stores", gb2, [&](int g) { hipLaunchKernelGGL((copy_kernel<4, true, true>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=8 nt loads + nt stores", gb2, [&](int g) { hipLaunchKernelGGL((copy_kernel<8, true, true>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=4 next trip requested before stores", gb2, [&](int g) { hipLaunchKernelGGL((copy_pipelined_kernel<4, false>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("copy U=4 pipelined, nt stores", gb2, [&](int g) { hipLaunchKernelGGL((copy_pipelined_kernel<4, true>), dim3(g), dim3(256), 0, 0, src, dst, n4); });
    row("read only U=4", gb1, [&](int g) { hipLaunchKernelGGL((read_kernel<4>), dim3(g), dim3(256), 0, 0, src, sink, n4); });
    row("read only U=8", gb1, [&](int g) { hipLaunchKernelGGL((read_kernel<8>), dim3(g), dim3(256), 0, 0, src, sink, n4); });
    row("write only U=4", gb1, [&](int g) { hipLaunchKernelGGL((fill_kernel<4, false>), dim3(g), dim3(256), 0, 0, dst, n4); });
    row("write only U=4 nt", gb1, [&](int g) { hipLaunchKernelGGL((fill_kernel<4, true>), dim3(g), dim3(256), 0, 0, dst, n4); });
    {
        printf("%-44s", "hipMemcpyAsync device to device");
        const double us = time_us([&] { CK(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, 0)); });
        printf("  %7.1f us %5.2f TB/s\n", us, gb2 / us * 1e3);
    }
    return 0;
}
