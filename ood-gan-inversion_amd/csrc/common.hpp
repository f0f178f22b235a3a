// Shared helpers for liboodgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <functional>
#include <utility>
#include "../../include/oodgan.h"

// ---- launch plans (include/oodgan.h, oodgan_plan_*; runtime.hip).  EVERY kernel launch of the library goes through launch_rec(): the
// launch — kernel, grid, block, dynamic LDS, stream and the by-value kernel arguments, i.e. the packed descriptor structs the entry
// points build — is a closure that is issued at once and, while the calling thread records a plan, also appended to it.  Replaying a
// plan re-issues the closures in order from C++: eager launches (no hipGraph), none of the host work that built them.
namespace oodgan {
struct PlanRec;
extern thread_local PlanRec* tl_plan_rec;
extern int g_null_launch;                  // != 0: closures skip the launch itself (host-cost probes: tools/plan_probe.py)
void plan_append(std::function<void()>&& f);
template <class F>
inline void launch_rec(F&& f) {
    f();
    if (__builtin_expect(tl_plan_rec != nullptr, 0)) plan_append(std::function<void()>(std::forward<F>(f)));
}
}  // namespace oodgan
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                        \
    ::oodgan::launch_rec([=]() {                                                                                                 \
        if (!::oodgan::g_null_launch) (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);      \
    })

namespace oodgan {

void set_error(const char* fmt, ...);

// dispatch tunables (runtime.hip): environment default read once, then oodgan_set_tunable
enum { OODGAN_TUN_S1_BIG_MIN_ITEMS = 0, OODGAN_TUN_S2_BIG_MIN_ITEMS, OODGAN_TUN_T2_BIG_MIN_ITEMS, OODGAN_TUN_BLURT_STRIP, OODGAN_TUN_BLUR_STRIP, OODGAN_TUN_UPVB_WAVES, OODGAN_TUN_FEWOUT_QUAD, OODGAN_TUN_TINY_MID_MAX, OODGAN_TUN_STRIPX_WAVES, OODGAN_TUN_COUNT };
long tunable(int id);

// dispatch counters (runtime.hip, oodgan_dispatch_count): which kernel family a conv call was routed to — tests assert that the
// kernel they mean to pin is the one that ran
enum { OODGAN_DC_STRIPX = 0, OODGAN_DC_STRIP, OODGAN_DC_S1BIG, OODGAN_DC_S1V2, OODGAN_DC_S1PP, OODGAN_DC_TINY, OODGAN_DC_T2BIG,
       OODGAN_DC_T2V2, OODGAN_DC_T2GEN, OODGAN_DC_S2BIG, OODGAN_DC_S2V2, OODGAN_DC_S2GEN, OODGAN_DC_UPVB,
       // sub-counters of the fused epilogues (round 5): 8-wave stride-1 conv writing ys / rgb partial sums; 8-wave stride-2 conv with the
       // fused activation backward; ... decoding its dotx from the saved S-form
       OODGAN_DC_S1BIG_YS, OODGAN_DC_S2BIG_FUSE, OODGAN_DC_S2BIG_DOTXS,
       // round 6: input-gradient launches with oodgan_conv_args.x_hi_only (two matrix instructions per product)
       OODGAN_DC_S1BIG_G2, OODGAN_DC_S2BIG_G2, OODGAN_DC_STRIPX_G2, OODGAN_DC_S2BIG_XH, OODGAN_DC_S1BIG_XH, OODGAN_DC_COUNT };
void count_dispatch(int id);

// One process per GPU: per-kernel setup (dynamic-LDS attributes, the zero page and CU count of the F-form strip
// conv) is done once per process, for the device that is current at first use.  bound_device_ok() records that device and makes
// every later conv call fail with OODGAN_E_ARG when another device is current, instead of launching with state of device 0.
bool bound_device_ok(const char* what);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return OODGAN_E_LAUNCH;
    }
    return OODGAN_OK;
}

#define OODGAN_REQUIRE(cond, ...)          \
    do {                                    \
        if (!(cond)) {                      \
            oodgan::set_error(__VA_ARGS__); \
            return OODGAN_E_ARG;            \
        }                                   \
    } while (0)

constexpr float kSqrt2 = 1.4142135623730951f;
constexpr float kInvPos = 1.f / kSqrt2, kInvNeg = 1.f / (0.2f * kSqrt2);      // inverse of lrelu(0.2) * sqrt2 on either branch
constexpr int kWave = 64;

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// grid size for a grid-stride streaming kernel: enough blocks to fill 256 CUs x 8, no more
inline int stream_grid(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Row sums of partials, shared by the per-layer and the batched kernels (they must agree bit for bit: tests/test_hip_generator.py).
// Loads are unconditional from a clamped index and masked by a select, eight (four) in flight per lane: `for (i = lane; i < n;
// i += 64) s += p[i]` is one round trip per trip.  n <= 64: 16 lanes per row (row_sum16; every 16-lane group of a wave may own a
// row of its own); longer rows: the whole wave.
__device__ __forceinline__ float row_sum16(const float* __restrict__ p, int n, int l) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p[min(l + 16 * u, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = l + 16 * u < n ? v[u] : 0.f;
    float r = (v[0] + v[1]) + (v[2] + v[3]);
    r += __shfl_xor(r, 8, 64);
    r += __shfl_xor(r, 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

__device__ __forceinline__ float row_sum_wave(const float* __restrict__ p, int n, int lane) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i0 = lane; i0 < n; i0 += 512) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[min(i0 + 64 * u, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + 64 * u < n ? v[u] : 0.f;
        a0 += v[0] + v[4]; a1 += v[1] + v[5]; a2 += v[2] + v[6]; a3 += v[3] + v[7];
    }
    return wave_sum((a0 + a1) + (a2 + a3));
}

// one row per wave, in the order the row's length selects (all lanes return the sum)
__device__ __forceinline__ float row_sum(const float* __restrict__ p, int n, int lane) {
    return n <= 64 ? row_sum16(p, n, lane & 15) : row_sum_wave(p, n, lane);
}

// sum_ci s[ci]^2 * wsq[ci] over a wave (demodulation, model.py:236-241): eight channel groups in flight per trip
__device__ __forceinline__ float demod_dot(const float* __restrict__ sp, const float* __restrict__ wp, int Ci, int lane) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int c0 = lane; c0 < Ci; c0 += 512) {
        float sv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ci = min(c0 + 64 * u, Ci - 1);
            sv[u] = sp[ci];
            wv[u] = wp[ci];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float pr = c0 + 64 * u < Ci ? sv[u] * sv[u] * wv[u] : 0.f;
            if ((u & 3) == 0) a0 += pr; else if ((u & 3) == 1) a1 += pr; else if ((u & 3) == 2) a2 += pr; else a3 += pr;
        }
    }
    return wave_sum((a0 + a1) + (a2 + a3));
}

// Streaming (non-temporal) accesses for tensors that are read or written ONCE per pass and are far larger than the caches (the
// activations / gradients of the >= 256² levels: 0.27-1.07 GB at batch 8).  tools/probes/copy_probe.hip, (8,32,1024,1024) fp32 read +
// write: plain 16-byte accesses 5.4-5.5 TB/s, `nt` loads and stores 5.7-5.9 TB/s (profiles/r4_copy_probe.txt).  -DOODGAN_NT_STREAM=0
// builds the plain forms (A/B).
#ifndef OODGAN_NT_STREAM
#define OODGAN_NT_STREAM 1
#endif
typedef float oodgan_f4v __attribute__((ext_vector_type(4)));
typedef float oodgan_f2v __attribute__((ext_vector_type(2)));
typedef unsigned oodgan_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
#if OODGAN_NT_STREAM
    const oodgan_f4v v = __builtin_nontemporal_load(reinterpret_cast<const oodgan_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ float2 ld2_stream(const float* p) {
#if OODGAN_NT_STREAM
    const oodgan_f2v v = __builtin_nontemporal_load(reinterpret_cast<const oodgan_f2v*>(p));
    return make_float2(v.x, v.y);
#else
    return *reinterpret_cast<const float2*>(p);
#endif
}
__device__ __forceinline__ float ld1_stream(const float* p) {
#if OODGAN_NT_STREAM
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
#if OODGAN_NT_STREAM
    __builtin_nontemporal_store(oodgan_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<oodgan_f4v*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
// a 16-byte slot of an S-form / phase-split record
template <typename T16>
__device__ __forceinline__ void st16_stream(void* p, const T16& v) {
    static_assert(sizeof(T16) == 16, "16-byte value");
#if OODGAN_NT_STREAM
    __builtin_nontemporal_store(__builtin_bit_cast(oodgan_u4v, v), reinterpret_cast<oodgan_u4v*>(p));
#else
    *reinterpret_cast<T16*>(p) = v;
#endif
}

// Sum over the 32 lanes of each half of the wave with DPP operands (VALU only): quad swaps, row_half_mirror, row_mirror and
// row_bcast:15 — the total of lanes 0-31 lands in lanes 16-31 (read it in lane 31), that of lanes 32-63 in lanes 48-63 (lane
// 63).  A __shfl_xor butterfly is five ds_bpermute_b32 per value: the fused epilogue of the stride-2 conv issued 486 of them
// per wave and tile, all eight waves of the CU at once on its one LDS pipe.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float half_sum_dpp(float v) {
    v += dpp_take<0xB1, 0xF>(v);         // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xF>(v);         // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xF>(v);        // row_half_mirror: the other quad of each 8
    v += dpp_take<0x140, 0xF>(v);        // row_mirror: the other half of each 16
    v += dpp_take<0x142, 0xA>(v);        // row_bcast:15 into rows 1 and 3 (rows 0 and 2 add the `old` operand, 0)
    return v;
}
constexpr int kHalfSumLane = 31;         // l31 of the lanes that hold half_sum_dpp's result

// sum over a 256-thread block; result valid in thread 0 (all threads must call)
__device__ __forceinline__ float block_sum_256(float v, float* red /*>=4 floats LDS*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace oodgan
