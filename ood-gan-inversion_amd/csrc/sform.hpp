// "S-form" activations: the layout the split-f16 conv kernels consume with zero conversion work.
//
//   S[b][kc = ceil(C/16)][Hp][Wp][slot 0..3][8 x f16]        64 bytes per pixel and 16-channel block
//     slot 0 = hi(ch 0-7)  slot 1 = hi(ch 8-15)  slot 2 = lo(ch 0-7)  slot 3 = lo(ch 8-15)
//   v = hi + lo (f16 split of the fp32 value, already multiplied by the consumer's per-(b,channel) scale:
//   style for the forward, demodulation x power-of-two range scale for the backward).
//   Pixel (y,x) of the image lives at [y+1][x+1]: a one-pixel ZERO border plus zero padding up to the conv
//   tile grid (Hp = roundup(H+1, 8) + 2, Wp = roundup(W+1, 32) + 2) so that a halo'd tile row is ONE
//   contiguous run (34 positions = 2176 B, ~95 % cache-line use; the NCHW fp32 tile rows were 160-B runs at
//   >=256-B stride, ~40 %) that the kernels fetch with LDS-DMA — no registers, no VALU, no bounds checks.
//   Producers write the interior only; the border is zeroed once at allocation and never touched.
#pragma once
#include "common.hpp"

namespace oodgan {

struct SDims {
    int C, H, W, KC, Hp, Wp;
    long plane;      // 16-byte units per (b,kc) plane
};

__host__ __device__ inline SDims sform_dims(int C, int H, int W) {
    SDims d;
    d.C = C; d.H = H; d.W = W;
    d.KC = (C + 15) / 16;
    d.Hp = (H + 1 + 7) / 8 * 8 + 2;
    d.Wp = (W + 1 + 31) / 32 * 32 + 2;
    d.plane = (long)d.Hp * d.Wp * 4;
    return d;
}

// index (16-byte units) of slot `s` of image pixel (y,x), channel block kc, batch b
__host__ __device__ inline long sform_unit(const SDims& d, int b, int kc, int y, int x, int s) {
    return (((long)b * d.KC + kc) * d.Hp + (y + 1)) * (long)d.Wp * 4 + (long)(x + 1) * 4 + s;
}

typedef _Float16 oodgan_half2v __attribute__((ext_vector_type(2)));

// hi/lo split of two values into packed f16 pairs.  The lo halves MUST be derived from the hi bits that are actually
// stored: the packed conversion (v_cvt_pk_f16_f32) and the scalar one the compiler would otherwise re-derive
// `(float)(_Float16)v` with do not agree on exact ties, which would leave hi and lo inconsistent (error 2^-11).
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
    oodgan_half2v h;
    h[0] = (_Float16)v0;
    h[1] = (_Float16)v1;
    hi = __builtin_bit_cast(unsigned, h);
    asm volatile("" : "+v"(hi));            // opaque: the halves below are read back from the packed register
    const oodgan_half2v hb = __builtin_bit_cast(oodgan_half2v, hi);
    oodgan_half2v l;
    l[0] = (_Float16)(v0 - (float)hb[0]);
    l[1] = (_Float16)(v1 - (float)hb[1]);
    lo = __builtin_bit_cast(unsigned, l);
}

// forward range control (fwd_range.hip): a producer records the max |v| of the values it wrote for sample b into one of
// OODGAN_VMAX_SLOTS slots of that sample (same-address atomics serialise in L2: 130k waves on 8 addresses cost
// milliseconds, spread over 512 addresses nothing).  Every lane of the wave must call, with the same b.
// Non-negative floats order like their bit patterns, and max is order independent: deterministic.
__device__ __forceinline__ void record_vmax(unsigned* vmax, int b, float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) {
        const unsigned slot = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (OODGAN_VMAX_SLOTS - 1);
        atomicMax(vmax + (long)b * OODGAN_VMAX_SLOTS + slot, __float_as_uint(m));
    }
}

// the same when the lanes of a wave may hold different samples (vb < 0: nothing recorded)
__device__ __forceinline__ void record_vmax_mixed(unsigned* vmax, int vb, float vm) {
    int bmax = vb, bmin = vb < 0 ? 0x7fffffff : vb;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bmax = max(bmax, __shfl_xor(bmax, o, 64));
        bmin = min(bmin, __shfl_xor(bmin, o, 64));
    }
    if (bmax < 0) return;
    if (bmin == bmax) record_vmax(vmax, bmax, vm);
    else if (vb >= 0 && vm > 0.f)
        atomicMax(vmax + (long)vb * OODGAN_VMAX_SLOTS + (threadIdx.x & (OODGAN_VMAX_SLOTS - 1)), __float_as_uint(vm));
}

}  // namespace oodgan
