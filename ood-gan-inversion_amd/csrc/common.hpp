// Shared helpers for liboodgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/oodgan.h"

namespace oodgan {

void set_error(const char* fmt, ...);

// dispatch tunables (runtime.hip): environment default read once, then oodgan_set_tunable
enum { OODGAN_TUN_S1_BIG_MIN_ITEMS = 0, OODGAN_TUN_S2_BIG_MIN_ITEMS, OODGAN_TUN_T2_BIG_MIN_ITEMS, OODGAN_TUN_BLURT_STRIP, OODGAN_TUN_BLUR_STRIP, OODGAN_TUN_COUNT };
long tunable(int id);

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
