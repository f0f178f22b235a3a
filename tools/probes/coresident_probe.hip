// Go / no-go probe for VERDICT r3 item 1: can a matrix-bound workgroup that LEAVES ROOM on its CU (<= 80 KB of LDS, 4 waves, <= 128
// arch VGPRs + accumulators) share the CU with the workgroups of a copy-bound kernel launched on another HIP stream — and is the pair
// then faster than the two kernels back to back?  Stand-alone, synthetic (no torch, no library): the matrix side is the tap loop of the
// 8-wave split-f16 conv (every fragment re-read from LDS by ds_read_b128, v_mfma_f32_16x16x32_f16, reads of step i+1 under the MFMAs of
// step i) with a per-tile prologue (a global -> LDS stage fill) and epilogue (64-byte stores); the copy side is a read+write stream of the
// (8,32,1024,1024) fp32 tensor with 4 x 16 bytes in flight per lane (tools/probes/copy_probe.hip), with or without an LDS allocation of
// the size the strip-walking producers use.
//
// Matrix configurations (same total number of MFMAs in all of them):
//   M8   8 waves, 150 KB LDS, one workgroup per tile                       — today's conv_f16s_s1big structure (fills the CU)
//   M4   4 waves,  76 KB LDS, one workgroup per tile (two fit on a CU)
//   M4p  4 waves,  76 KB LDS, persistent: 256 workgroups (one per CU) walking tiles — half of every CU stays free for the other queue
//   M4q  4 waves,  76 KB LDS, persistent: 512 workgroups (two per CU)
// Output: each kernel alone, the pair on two streams, serial sum, gain = serial / pair.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/coresident_probe.hip -o tools/probes/build/coresident_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

struct MArgs {
    const uint4* src;     // >= 160 KB of f16 noise
    float* out;           // tile outputs: tiles x waves x 64 lanes x 16 floats
    int tiles;            // total tiles of the launch
    int steps;            // K steps per tile (each: 8 fragment reads + 16 MFMAs per wave)
    int stage_bytes;      // global -> LDS bytes per tile prologue (one stage fill)
};

// One tile of one wave: prologue (workgroup-wide stage fill + barrier), `steps` x {8 ds_read_b128, 16 MFMA 16x16x32}, epilogue stores.
template <int WAVES>
__device__ __forceinline__ void tile_body(const MArgs& p, unsigned char* smem, int tile, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    // stage fill: 16 bytes per thread and trip, contiguous
    for (int i = tid * 16; i < p.stage_bytes; i += WAVES * 64 * 16)
        *reinterpret_cast<uint4*>(smem + i) = p.src[((long)tile * 97 + (i >> 4)) & 8191];
    __syncthreads();
    const unsigned char* lw = smem + 40 * 1024 + lane * 16;
    const unsigned char* lx = smem + (wave & 3) * 2176 + lane * 16;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct F { half8 a[4], b[4]; };
    auto load = [&](F& f, int t) {
        const int o = (t % 9) * 2048;
#pragma unroll
        for (int m = 0; m < 4; ++m) f.a[m] = *reinterpret_cast<const half8*>(lw + o + m * 1024 * 4);
#pragma unroll
        for (int n = 0; n < 4; ++n) f.b[n] = *reinterpret_cast<const half8*>(lx + (t % 3) * 64 + n * 8704);
    };
    auto mfma = [&](const F& f) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[m], f.b[n], acc[m][n], 0, 0, 0);
    };
    F f0, f1;
    load(f0, 0);
    SB();
    for (int t = 0; t < p.steps; t += 2) {
        load(f1, t + 1); SB(); mfma(f0); SB();
        load(f0, t + 2); SB(); mfma(f1); SB();
    }
    float* o = p.out + (((long)tile * WAVES + wave) * 64 + lane) * 64;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(o + (a * 4 + b) * 4) = acc[a][b];
    __syncthreads();
}

template <int WAVES, bool PERSISTENT>
__global__ __launch_bounds__(WAVES * 64) void matrix_kernel(const MArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (PERSISTENT) {
        for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) tile_body<WAVES>(p, smem, tile, threadIdx.x);
    } else {
        tile_body<WAVES>(p, smem, blockIdx.x, threadIdx.x);
    }
}

// copy side: U = 4 independent 16-byte loads per lane and trip; LDSB bytes of (unused but allocated) LDS per workgroup
template <int LDSB>
__global__ __launch_bounds__(256) void copy_kernel(const float4v* __restrict__ src, float4v* __restrict__ dst, long n4) {
    __shared__ float pad[LDSB > 0 ? LDSB / 4 : 1];
    if (LDSB > 0 && n4 < 0) pad[threadIdx.x] = 1.f;       // keep the allocation
    constexpr int U = 4;
    const long stride = (long)gridDim.x * 256 * U;
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
        float4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[std::min(base + u * 256, n4 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (base + u * 256 < n4) dst[base + u * 256] = v[u];
    }
    if (LDSB > 0 && n4 < 0) dst[0].x = pad[0];
}

static hipStream_t sa, sb;

template <typename FA, typename FB>
static double pair_us(FA fa, FB fb, bool run_a, bool run_b, int n = 10) {
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    std::vector<double> ts;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipDeviceSynchronize());
        if (run_a) CK(hipEventRecord(a0, sa));
        if (run_b) CK(hipEventRecord(b0, sb));
        for (int i = 0; i < n; ++i) {
            if (run_a) fa();
            if (run_b) fb();
        }
        if (run_a) CK(hipEventRecord(a1, sa));
        if (run_b) CK(hipEventRecord(b1, sb));
        CK(hipDeviceSynchronize());
        float ta = 0.f, tb = 0.f;
        if (run_a) CK(hipEventElapsedTime(&ta, a0, a1));
        if (run_b) CK(hipEventElapsedTime(&tb, b0, b1));
        ts.push_back(std::max(ta, tb) * 1e3 / n);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const long n = 8L * 32 * 1024 * 1024, n4 = n / 4;
    float4v *csrc, *cdst;
    CK(hipMalloc(&csrc, n * 4));
    CK(hipMalloc(&cdst, n * 4));
    CK(hipMemset(csrc, 0x11, n * 4));
    uint4* msrc;
    CK(hipMalloc(&msrc, 8192 * 16));
    std::vector<_Float16> h(8192 * 8);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 20) - 2048) * (0.05f / 2048)); }
    CK(hipMemcpy(msrc, h.data(), 8192 * 16, hipMemcpyHostToDevice));
    // 8-wave tiles: TILES8 tiles x 8 waves x steps; 4-wave tiles: twice as many tiles (same MFMAs, same stores)
    const int TILES8 = 2048, STEPS = 36 * 4;             // K = 64: four 16-channel chunks x 9 taps x {hi.hi, hi.lo, lo.hi} / pair ~ 36 steps per chunk
    float* mout;
    CK(hipMalloc(&mout, (size_t)TILES8 * 8 * 64 * 64 * 4));
    const int L8 = 150 * 1024, L4 = 76 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&matrix_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, L8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&matrix_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, L4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&matrix_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, L4));
    MArgs a8{msrc, mout, TILES8, STEPS, 75 * 1024}, a4{msrc, mout, TILES8 * 2, STEPS, 38 * 1024};
    auto m8 = [&] { hipLaunchKernelGGL((matrix_kernel<8, false>), dim3(a8.tiles), dim3(512), L8, sa, a8); };
    auto m4 = [&] { hipLaunchKernelGGL((matrix_kernel<4, false>), dim3(a4.tiles), dim3(256), L4, sa, a4); };
    auto m4p = [&] { hipLaunchKernelGGL((matrix_kernel<4, true>), dim3(256), dim3(256), L4, sa, a4); };
    auto m4q = [&] { hipLaunchKernelGGL((matrix_kernel<4, true>), dim3(512), dim3(256), L4, sa, a4); };
    auto c0 = [&] { hipLaunchKernelGGL((copy_kernel<0>), dim3(2048), dim3(256), 0, sb, csrc, cdst, n4); };
    auto c19 = [&] { hipLaunchKernelGGL((copy_kernel<19 * 1024>), dim3(2048), dim3(256), 0, sb, csrc, cdst, n4); };
    const double flop = (double)TILES8 * 8 * STEPS * 16 * 2.0 * 16 * 16 * 32;
    printf("matrix side: %.1f GFLOP of MFMA per launch; copy side: %.2f GB read + %.2f GB written per launch\n", flop / 1e9, n * 4 / 1e9, n * 4 / 1e9);
    auto report = [&](const char* mn, auto mf, const char* cn, auto cf) {
        const double tm = pair_us(mf, cf, true, false), tc = pair_us(mf, cf, false, true), tp = pair_us(mf, cf, true, true);
        printf("%-5s + %-22s: matrix alone %7.1f us (%5.0f TF/s)  copy alone %7.1f us (%4.2f TB/s)  pair %7.1f us  serial %7.1f  gain %.3f\n", mn, cn, tm,
               flop / tm / 1e6, tc, 2.0 * n * 4 / tc / 1e6, tp, tm + tc, (tm + tc) / tp);
        fflush(stdout);
    };
    for (int round = 0; round < 2; ++round) {
        report("M8", m8, "copy (no LDS)", c0);
        report("M4", m4, "copy (no LDS)", c0);
        report("M4p", m4p, "copy (no LDS)", c0);
        report("M4q", m4q, "copy (no LDS)", c0);
        report("M8", m8, "copy (19 KB LDS)", c19);
        report("M4", m4, "copy (19 KB LDS)", c19);
        report("M4p", m4p, "copy (19 KB LDS)", c19);
        report("M4q", m4q, "copy (19 KB LDS)", c19);
    }
    return 0;
}
