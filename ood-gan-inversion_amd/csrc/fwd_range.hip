// Forward range control of the split-f16 path.
//
// An S-form record holds v = x[b,c,p] * s[b,c] as hi + lo f16: |v| above 65504 would overflow, far below 1 the lo half
// runs out of f16 subnormals.  Every styled conv l therefore carries one power-of-two scale per sample, q[l][b], chosen
// so that max_{c,p} |v| * q lies in [512,1024).  It costs no kernel work: the producers are simply handed the style
// column block s'[b,c] = s[b,c]*q and the conv epilogue the demodulation block d'[b,m] = d[b,m]/q — both exact.
//   oodgan_absmax_scaled      vmax[b] = max |x*s|              (exact mode: measured before the tensor is converted)
//   oodgan_fwd_range_update   vmax -> (flag, next q), vmax = 0  (carry mode: the producers recorded max |v*q_used|)
//   oodgan_fwd_range_plan     s_sc = s_all*q[layer(row)], d_sc = d_all/q[layer(drow)]
// Reference semantics kept: ModulatedConv2d.forward (src/ops/StyleGAN/model.py:233-274) in fp32 has no such limit.
#include "common.hpp"
#include <cstdint>

using namespace oodgan;

namespace {

// grid (nblk, C, B): block max of |x[b,c,:]| over its chunk, times |s[b,c]|, atomically maxed into a slot of vmax[b][:]
// (non-negative floats order like their bit patterns, and max is order independent: deterministic)
__global__ __launch_bounds__(256) void absmax_scaled_kernel(const float* __restrict__ x, const float* __restrict__ s, int s_stride,
                                                            unsigned* __restrict__ vmax, int C, long HW, long chunk) {
    __shared__ float red[4];
    const int b = blockIdx.z, c = blockIdx.y;
    const float* xp = x + ((long)b * C + c) * HW;
    const long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < HW ? p0 + chunk : HW;
    float m = 0.f;
    if ((HW & 3) == 0 && (chunk & 3) == 0) {
        for (long p = p0 + 4L * threadIdx.x; p < p1; p += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(xp + p);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            if (!(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w))) m = INFINITY;
        }
    } else {
        for (long p = p0 + threadIdx.x; p < p1; p += 256) {
            const float v = xp[p];
            m = isfinite(v) ? fmaxf(m, fabsf(v)) : INFINITY;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * fabsf(s ? s[(long)b * s_stride + c] : 1.f);
        if (!(m == m)) m = INFINITY;          // 0 * inf
        if (m > 0.f) atomicMax(vmax + (long)b * OODGAN_VMAX_SLOTS + ((blockIdx.x + blockIdx.y) & (OODGAN_VMAX_SLOTS - 1)), __float_as_uint(m));
    }
}

// one thread per (layer, sample) entry
__global__ void fwd_range_update_kernel(unsigned* __restrict__ vmax, float* __restrict__ q, int* __restrict__ flag, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float m = 0.f;
    for (int k = 0; k < OODGAN_VMAX_SLOTS; ++k) {
        m = fmaxf(m, __uint_as_float(vmax[(long)i * OODGAN_VMAX_SLOTS + k]));
        vmax[(long)i * OODGAN_VMAX_SLOTS + k] = 0u;
    }
    float t = m;                            // true max |x*s|
    if (flag) {                             // carry mode: m was measured on values scaled by q[i]
        if (!isfinite(m)) atomicOr(flag, 2);
        // window of a carried scale: [1, 2^15).  Round 5: the lower bound was 2^-8 — there the lo halves of the values near the maximum
        // are f16 subnormals (16 significant bits relative to the maximum instead of 22); from 1 on they are normal numbers.  The scale aims at
        // [512, 1024), so a carried scale survives a 512-fold drop and a 32-fold rise of the maximum between two forwards
        else if (m > 0.f && !(m >= 1.f && m < 32768.f)) atomicOr(flag, 1);
        t = m / q[i];
    }
    int e = 0;
    if (t > 0.f && isfinite(t)) e = 9 - (int)floorf(log2f(t));
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    q[i] = ldexpf(1.f, e);
}

// element (b, j): j < nrows -> style row row0+j, else demodulation row drow0 + (j-nrows)
__global__ __launch_bounds__(256) void fwd_range_plan_kernel(const float* __restrict__ s_all, const float* __restrict__ d_all,
                                                             const int* __restrict__ row_layer, const int* __restrict__ drow_layer,
                                                             const float* __restrict__ q, float* __restrict__ s_sc,
                                                             float* __restrict__ d_sc, int B, int R, int DR, int row0, int nrows,
                                                             int drow0, int ndrows) {
    const int per = nrows + ndrows;
    const long total = (long)B * per;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int b = (int)(e / per), j = (int)(e % per);
        if (j < nrows) {
            const int r = row0 + j, l = row_layer[r];
            const float v = s_all[(long)b * R + r];
            s_sc[(long)b * R + r] = l >= 0 ? v * q[(long)l * B + b] : v;
        } else {
            const int r = drow0 + (j - nrows), l = drow_layer[r];
            const float v = d_all[(long)b * DR + r];
            d_sc[(long)b * DR + r] = l >= 0 ? v / q[(long)l * B + b] : v;
        }
    }
}

}  // namespace

extern "C" int oodgan_absmax_scaled(const float* x, const float* s, int s_stride, unsigned* vmax, int B, int C, long HW,
                                    void* stream) {
    OODGAN_REQUIRE(x && vmax && B > 0 && C > 0 && HW > 0, "absmax_scaled: bad args");
    OODGAN_REQUIRE(B <= 65535 && C <= 65535, "absmax_scaled: B or C too large");
    if (s == nullptr && C > 1) {
        // no per-channel scale: one flat range of C*HW values per sample instead of a block per (b, c) plane — the stacked style heads
        // (9216 channels of 16 x 16 and below) were 73 728 blocks of 256 values each: 132 us for 75 MB
        HW *= C;
        C = 1;
    }
    long chunk = 16384;                       // elements per block
    long nblk = (HW + chunk - 1) / chunk;
    if (nblk > 1024) { nblk = 1024; chunk = ((HW + nblk - 1) / nblk + 3) / 4 * 4; nblk = (HW + chunk - 1) / chunk; }
    hipLaunchKernelGGL(absmax_scaled_kernel, dim3((unsigned)nblk, C, B), dim3(256), 0, as_stream(stream), x, s, s_stride, vmax, C,
                       HW, chunk);
    return check_launch("absmax_scaled");
}

extern "C" int oodgan_fwd_range_update(unsigned* vmax, float* q, int* flag, int n, void* stream) {
    OODGAN_REQUIRE(vmax && q && n > 0, "fwd_range_update: bad args");
    hipLaunchKernelGGL(fwd_range_update_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), vmax, q, flag, n);
    return check_launch("fwd_range_update");
}

extern "C" int oodgan_fwd_range_plan(const float* s_all, const float* d_all, const int* row_layer, const int* drow_layer,
                                     const float* q, float* s_sc, float* d_sc, int B, int R, int DR, int row0, int nrows, int drow0,
                                     int ndrows, void* stream) {
    OODGAN_REQUIRE(s_all && d_all && row_layer && drow_layer && q && s_sc && d_sc && B > 0, "fwd_range_plan: bad args");
    OODGAN_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= R && drow0 >= 0 && ndrows >= 0 && drow0 + ndrows <= DR && nrows + ndrows > 0,
                   "fwd_range_plan: row range");
    const long total = (long)B * (nrows + ndrows);
    hipLaunchKernelGGL(fwd_range_plan_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), s_all, d_all, row_layer,
                       drow_layer, q, s_sc, d_sc, B, R, DR, row0, nrows, drow0, ndrows);
    return check_launch("fwd_range_plan");
}
