// Shared pieces of the implicit-GEMM conv kernels (fp32-MFMA and split-f16-MFMA variants):
// launch geometry, XCD-aware work mapping and the epilogue (identical C/D register layout for
// every 32x32 MFMA shape on gfx950).
#pragma once
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace oodgan {

struct KArgs {
    oodgan_conv_args a;
    int Hn, Wn;          // N-space size
    int Hout, Wout;      // output size
    int tiles_x, tiles_y;
    int mblocks, Mp;     // Mp = padded M of the packed weights
    long in_plane, out_plane;
    const float* w_unscale;   // device scalar multiplying every accumulator (split-f16 weights are pre-scaled); NULL = 1
    int ablate;               // debug: bit0 skip MFMA, bit1 skip x loads, bit2 skip w loads, bit3 skip LDS stores, bit4 skip epilogue
};

__device__ __forceinline__ int xcd_remap(int bid, int total) {
    // contiguous chunk of the work list per XCD (blocks b and b+8 share an XCD): neighbouring
    // work items (same weight block, adjacent tiles) hit the same L2.  Bijective for any total.
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = total >> 3, r = total & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct BlockCtx {
    int tile, b, mblk, r0, c0, m0;
};

template <int TR, int MB>
__device__ __forceinline__ BlockCtx decode_block(const KArgs& p) {
    BlockCtx c;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = p.tiles_x * p.tiles_y;
    c.tile = w % ntile;
    w /= ntile;
    c.b = w % p.a.B;
    c.mblk = w / p.a.B;
    c.r0 = (c.tile / p.tiles_x) * TR;
    c.c0 = (c.tile % p.tiles_x) * 32;
    c.m0 = c.mblk * MB;
    return c;
}

// acc[mt][j]: j = N-tile (S1/S2) or output phase py*2+px (T2)
template <int MODE, int MT, int NT, int NACC>
__device__ __forceinline__ void conv_epilogue(const KArgs& p, f32x16 (&acc)[MT][NACC], const BlockCtx& c, int wave, int l31,
                                              int half) {
    const oodgan_conv_args& a = p.a;
    const int b = c.b, m0 = c.m0, r0 = c.r0, c0 = c.c0;
    const float* osc = a.out_scale ? a.out_scale + (long)b * a.out_scale_stride : nullptr;
    float* yb = a.y + (long)b * a.M * p.out_plane;
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * ((p.w_unscale && a.in_mul2) ? a.in_mul2[0] : 1.f);
    if constexpr (MODE == OODGAN_CONV_T2) {
        const int ip = r0 + wave;          // i'
        const int jp = c0 + l31;           // j'
        const int zx = 2 * jp;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // the 16 scales of the M-tile from clamped channels, all in flight together: a load inside the store loop is a branch, a
            // load and a vmcnt(0) per value, which also waits for the two stores in front of it
            float scv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = min(m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.M - 1);
                scv[r] = osc ? osc[m] : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= a.M || zx >= p.Wout) continue;
                const float sc = scv[r] * us;
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int zy = 2 * ip + py;
                    if (zy >= p.Hout) continue;
                    float2 v = make_float2(acc[mt][py * 2 + 0][r] * sc, acc[mt][py * 2 + 1][r] * sc);
                    // out_pitch is even and out_plane is even for T2 (host guarantees) -> 8-byte aligned
                    *reinterpret_cast<float2*>(yb + (long)m * p.out_plane + (long)zy * a.out_pitch + zx) = v;
                }
            }
        }
    } else {
    const int px = c0 + l31;
    const float* db = a.dotx ? a.dotx + (long)b * a.M * ((long)p.Hout * p.Wout) : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float dsum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dsum[r] = 0.f;
        // the per-channel constants of the M-tile, the noise and the dot operand of a row: unconditional loads from clamped channels /
        // pixels (`yb`: a valid address for an absent tensor), all of a row in flight together.  `if (bias) v += bias[m]` inside the
        // store loop is a branch, a load and a vmcnt(0) per value — up to 64 serialised round trips per M-tile, each waiting for the
        // store before it
        const float* oscp = osc ? osc : yb;
        const float* biasp = a.bias ? a.bias : yb;
        const float* slopep = (a.act == OODGAN_ACT_PRELU && a.slope) ? a.slope : yb;
        float oscv[16], biav[16], slpv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mc = min(m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.M - 1);
            oscv[r] = oscp[mc];
            biav[r] = biasp[mc];
            slpv[r] = slopep[mc];
        }
        const long HWo = (long)p.Hout * p.Wout;
        const float* nzp = a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * HWo : yb;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int py = r0 + wave * NT + nt;
            const bool pix_ok = (py < p.Hout) && (px < p.Wout);
            const long pixc = (long)min(py, p.Hout - 1) * p.Wout + min(px, p.Wout - 1);
            const float nzr = nzp[a.noise ? pixc : 0];
            float dv[16];
            if (db) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mc = min(m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, a.M - 1);
                    dv[r] = db[(long)mc * HWo + pixc];
                }
            }
            const float nz = (a.noise && pix_ok) ? nw * nzr : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (!pix_ok || m >= a.M) continue;
                float v = acc[mt][nt][r] * us;
                if (db) dsum[r] += v * dv[r];
                if (osc) v *= oscv[r];
                v += nz;
                if (a.bias) v += biav[r];
                if (a.act == OODGAN_ACT_LRELU) v = (v > 0.f ? v : 0.2f * v) * kSqrt2;
                else if (a.act == OODGAN_ACT_PRELU) v = v > 0.f ? v : slpv[r] * v;
                yb[(long)m * p.out_plane + (long)py * a.out_pitch + px] = v;
            }
        }
        if (db) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s = half_sum_dpp(dsum[r]);
                const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (l31 == kHalfSumLane && m < a.M)
                    a.dot_part[((long)b * a.M + m) * a.dot_nparts + c.tile * 4 + wave] = s;
            }
        }
    }
    }
}

}  // namespace oodgan
