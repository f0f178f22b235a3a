// Shared helpers for liboodgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include "../../include/oodgan.h"

namespace oodgan {

void set_error(const char* fmt, ...);

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
