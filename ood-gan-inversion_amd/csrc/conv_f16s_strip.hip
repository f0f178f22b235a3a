// Split-f16 3x3 stride-1 conv for the 32 -> 32 channel layers at full resolution (the 1024² ModulatedConv2d of
// reference src/ops/StyleGAN/model.py:233-274 — forward — and its input gradient), the HBM-bound end of the path.
// Same arithmetic as conv_f16s_s1v2_kernel (S-form input, hi/lo split, 3 MFMAs per product, fp32 accumulate), but
// organised like modconv_f16_strip_kernel (conv_f16.hip) so that HBM, not latency, sets the time:
//   * the whole weight tensor (9 taps x 2 channel blocks x {hi,lo}) lives in REGISTERS (36 A fragments);
//   * a workgroup walks DOWN a 32-pixel-wide strip; the input rows live in a 32-row LDS ring (4 groups of 8 rows,
//     fetched two groups ahead by LDS-DMA), so every input row is fetched once per strip;
//   * S-form records are 64 bytes; the four 16-byte slots of record c are stored rotated by (c>>2)&3 — done for free
//     by the per-lane SOURCE address of the DMA — which makes the ds_read_b128 fragment reads bank-conflict free
//     without the 80-byte padded records of the tile kernels;
//   * the epilogue runs from the accumulators (three independent chains per output row: hi*hi, hi*lo, lo*hi) and
//     stores fp32 rows of 128 contiguous bytes per half wave; the backward instance multiplies by the saved forward
//     input for the style gradient (dot epilogue) and reduces it across the workgroup through LDS.
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>
#include <cstdlib>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

#ifndef STRIP_ABL
#define STRIP_ABL 0
#endif

namespace {

constexpr int SC_C = 34;                               // records per ring row and channel block
constexpr int SC_ROW = 2 * SC_C * 64;                  // 4352 bytes: [kc 2][34 records][64 B]
constexpr int SC_GROUP = 8 * SC_ROW;                   // 34816 bytes = 34 one-KiB DMA pieces
constexpr int SC_PIECES = SC_GROUP / 1024;             // 34
constexpr int SC_RING = 32 * SC_ROW;                   // 139264
constexpr int SC_NOISE = SC_RING;                      // 4 x 1 KiB noise tiles
constexpr int SC_RED = SC_NOISE + 4096;                // 2 x 4 waves x 32 floats: dot partials of two tiles
constexpr int SC_SMEM = SC_RED + 2 * 4 * 32 * 4;

struct StripConv {
    oodgan_conv_args a;
    const uint4* xs;
    SDims xd;
    const float* w_unscale;
    int tiles_x, tiles_y, seg_tiles, nseg, Mp;
    long out_plane;
    int counted_wait;
};

// PRE (input-gradient instance, oodgan_conv_args.dot_actgrad): as in conv_f16s_s1big_kernel — y <- dx * act'(dotx).
template <bool DOT, bool RGB, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_f16s_strip_kernel(
    const StripConv p, const uint4* __restrict__ wpk16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    int w = blockIdx.x;
    {
        const int total = gridDim.x, xcd = w & 7, idx = w >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = w % p.tiles_x;
    const int seg = (w / p.tiles_x) % p.nseg;
    const int b = w / (p.tiles_x * p.nseg);
    const int t0 = seg * p.seg_tiles;
    const int n = min(p.seg_tiles, p.tiles_y - t0);
    const int c0 = tx * 32, R0 = 8 * t0;
    const int H = a.Hin, W = a.Win, M = a.M;

    // ---- weights: the whole tensor in registers.  Packed order (oodgan_pack_conv3x3_f16s):
    // [kc][tap][hi|lo][k-half][Mp][8 f16]
    half8 ah[9][2], al[9][2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const half8* wb = reinterpret_cast<const half8*>(wpk16) + ((long)(kc * 9 + tp) * 4) * p.Mp;
            ah[tp][kc] = wb[(0 * 2 + half) * p.Mp + l31];
            al[tp][kc] = wb[(1 * 2 + half) * p.Mp + l31];
        }

    // ---- epilogue constants of this lane's 16 output channels
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
    float osc[16], bia[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        osc[r] = (m < M ? (a.out_scale ? a.out_scale[(long)b * a.out_scale_stride + m] : 1.f) : 0.f) * us;
        bia[r] = (!DOT && a.bias && m < M) ? a.bias[m] : 0.f;
    }
    // fused ToRGB: this lane's 16 channels of the three modulated 1x1 rows (rgb_scale * w[k,m] * s_rgb[b,m])
    float wr[RGB ? 3 : 1][16];
    if (RGB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float sv = m < M ? a.rgb_scale * a.rgb_s[(long)b * a.rgb_s_stride + m] : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) wr[RGB ? k : 0][r] = m < M ? sv * a.rgb_w[k * M + m] : 0.f;
        }
    }
    const float nw = (!DOT && a.noise) ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    // byte offsets of this lane's 16 channel planes (32-bit: M*plane*4 < 4 GiB is checked by the host) + its column
    unsigned moff[16], doff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        moff[r] = (unsigned)((long)m * p.out_plane * 4) + l31 * 4;
        doff[r] = (unsigned)((long)m * H * W * 4) + l31 * 4;
    }
    const bool mfull = M == 32;
    const float* nzb = (a.noise ? a.noise : reinterpret_cast<const float*>(p.xs)) + (long)(a.noise_batch > 1 ? b : 0) * H * W;

    // ---- per-lane DMA source offsets inside a group (bytes from the group's first row at column c0); the slot
    // rotation (c>>2)&3 of the LDS image is applied here
    unsigned xoff[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int P = (wave + 4 * i) * 64 + lane;
        if (P >= SC_GROUP / 16) P = SC_GROUP / 16 - 1;
        const int row = P / 272, q = P % 272;
        const int kc = q / 136, q2 = q % 136;
        const int c = q2 >> 2, s = ((q2 & 3) - ((c >> 2) & 3)) & 3;
        xoff[i] = (unsigned)(((long)kc * p.xd.plane + ((long)row * p.xd.Wp + c) * 4 + s) * 16);
    }
    const int npc = wave < 2 ? 9 : 8;                       // 34 pieces over 4 waves
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xs) + ((long)b * 2 * p.xd.plane + (long)c0 * 4) * 16;
    const long row_bytes = (long)p.xd.Wp * 64;

    auto dma_group = [&](int g) {
        const int gg = min(g, n - 1);                        // past the end: re-fetch the last group into a dead ring group
        const unsigned char* base = xb + (long)(R0 + 2 + 8 * gg) * row_bytes;
        unsigned char* dst = smem + (g & 3) * SC_GROUP;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xoff[i]),
                                             (lds_void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
        if (wave < 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xoff[8]),
                                             (lds_void*)(dst + (wave + 32) * 1024), 16, 0, 0);
        if (!DOT) {   // noise of tile g: each wave its own two rows, one dword per lane, clamped inside the image
            const int ny = min(R0 + 8 * gg + wave * 2 + half, H - 1), nx = min(c0 + l31, W - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nzb + (long)ny * W + nx),
                                             (lds_void*)(smem + SC_NOISE + (g & 3) * 1024 + wave * 256), 4, 0, 0);
        }
    };

    constexpr int kVm0 = 0x0F70;
    __builtin_amdgcn_s_waitcnt(kVm0);            // weights, scales: retired here, never inside the loop
    {   // prologue: the two halo rows above the first tile (tail of ring group 3), then groups 0 and 1
        const int Rm = R0 + 2 - 8;
        unsigned char* dst = smem + 3 * SC_GROUP;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (i == 8 && wave >= 2) break;
            int P = (wave + 4 * i) * 64 + lane;
            if (P >= SC_GROUP / 16) P = SC_GROUP / 16 - 1;
            const int row = P / 272;
            const int rr = max(Rm + row, 0) - (Rm + row);    // rows above the image -> row 0 (never used)
            const unsigned char* src = xb + (long)(Rm + rr) * row_bytes + xoff[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (lds_void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
        }
        dma_group(0);
        dma_group(1);
    }

    // lane-constant part of the fragment addresses: record kx + l31, slot (half + 2*lo) rotated by (c>>2)&3
    unsigned lrd[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
            const int c = kx + l31;
            lrd[kx][lo] = c * 64 + (((half + 2 * lo + ((c >> 2) & 3)) & 3) << 4);
        }
    const int px = c0 + l31;
    const bool col_full = c0 + 32 <= W && M == 32 && p.counted_wait;
    int ragged = 3;
    float* red = reinterpret_cast<float*>(smem + SC_RED);

    for (int t = 0; t < n; ++t) {
        // group t must have landed.  Memory operations retire in issue order.  Forward: behind group t there may be
        // group t+1 with its noise (npc+1) and the 32 stores of tile t-1 (the stores of tile t-2 are also waited for:
        // the counter only has 6 bits).  Backward: the epilogue's dot loads were issued after the prefetch, so
        // consuming them already implied that every earlier operation had completed.
        if (!DOT) {
            if (t == 0) {
                if (wave < 2) __builtin_amdgcn_s_waitcnt(0x0F7A); else __builtin_amdgcn_s_waitcnt(0x0F79);       // 10 / 9
            } else if (ragged & 1) {
                __builtin_amdgcn_s_waitcnt(kVm0);
            } else {
                // allowed in flight: the stores of tile t-1 and group t+1 with its noise (10 / 9 operations)
                if (a.y_fform) {        // 8 float4 stores per tile (+ 6 colour stores)
                    if (RGB) {
                        if (wave < 2) __builtin_amdgcn_s_waitcnt(0x4F78); else __builtin_amdgcn_s_waitcnt(0x4F77);   // 24 / 23
                    } else {
                        if (wave < 2) __builtin_amdgcn_s_waitcnt(0x4F72); else __builtin_amdgcn_s_waitcnt(0x4F71);   // 18 / 17
                    }
                } else if (RGB) {       // 32 dword stores + 6 colour stores
                    if (wave < 2) __builtin_amdgcn_s_waitcnt(0xCF70); else __builtin_amdgcn_s_waitcnt(0x8F7F);   // 48 / 47
                } else {
                    if (wave < 2) __builtin_amdgcn_s_waitcnt(0x8F7A); else __builtin_amdgcn_s_waitcnt(0x8F79);   // 42 / 41
                }
            }
        } else if (t == 0) {
            if (wave < 2) __builtin_amdgcn_s_waitcnt(0x0F79); else __builtin_amdgcn_s_waitcnt(0x0F78);           // 9 / 8
        }
        __builtin_amdgcn_s_barrier();
        if (!(STRIP_ABL & 8)) dma_group(t + 2);
        const int ty = t0 + t;
        const bool full = col_full && R0 + 8 * t + 8 <= H;
        ragged = ((ragged << 1) | (full ? 0 : 1)) & 3;

        // backward: the saved forward input of this tile (for the style gradient); forward: noise from LDS
        float dxv[2][16];
        if (DOT) {
            const unsigned char* db = reinterpret_cast<const unsigned char*>(a.dotx) +
                                      ((long)b * M * H * W + (long)(R0 + 8 * t + wave * 2) * W + c0) * 4;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int py = R0 + 8 * t + wave * 2 + nt;
                const bool ok = py < H && px < W;
#pragma unroll
                for (int r = 0; r < 16; ++r) dxv[nt][r] = 0.f;
                if (STRIP_ABL & 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dxv[nt][r] = 1.f + r;
                } else if (ok) {
                    if (mfull) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dxv[nt][r] = *reinterpret_cast<const float*>(db + (long)nt * W * 4 + doff[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if ((r & 3) + 8 * (r >> 2) + 4 * half < M) dxv[nt][r] = *reinterpret_cast<const float*>(db + (long)nt * W * 4 + doff[r]);
                    }
                }
            }
            if (t > 0 && wave == 0 && lane < 32 && lane < M) {
                // cross-wave sum of the previous tile's dot partials (written to LDS before the barrier above)
                const float* rp = red + ((t - 1) & 1) * 128;
                a.dot_part[((long)b * M + lane) * a.dot_nparts + (long)(ty - 1) * p.tiles_x + tx] =
                    rp[lane] + rp[32 + lane] + rp[64 + lane] + rp[96 + lane];
            }

        }

        f32x16 acc[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][k][r] = 0.f;
        unsigned rbase[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) rbase[rr] = ((8 * t + 30 + 2 * wave + rr) & 31) * SC_ROW;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                half8 bh[2][3], bl[2][3];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const unsigned char* base = smem + rbase[nt + ky] + kc * (SC_C * 64);
                        if (STRIP_ABL & 16) {
                            bh[nt][kx] = ah[kx][nt];
                            bl[nt][kx] = al[kx][nt];
                        } else {
                            bh[nt][kx] = *reinterpret_cast<const half8*>(base + lrd[kx][0]);
                            bl[nt][kx] = *reinterpret_cast<const half8*>(base + lrd[kx][1]);
                        }
                    }
                if (STRIP_ABL & 4) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int tp = ky * 3 + kx;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tp][kc], bh[nt][kx], acc[nt][0], 0, 0, 0);
                        acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tp][kc], bl[nt][kx], acc[nt][1], 0, 0, 0);
                        acc[nt][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tp][kc], bh[nt][kx], acc[nt][2], 0, 0, 0);
                    }
                }
            }
        }
        // ---- epilogue
        unsigned char* ybt = reinterpret_cast<unsigned char*>(a.y) +
                             ((long)b * M * p.out_plane + (long)(R0 + 8 * t + wave * 2) * a.out_pitch + c0) * 4;
        float dsum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dsum[r] = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int py = R0 + 8 * t + wave * 2 + nt;
            const bool ok = py < H && px < W;
            float nz = 0.f;
            if (!DOT) nz = nw * reinterpret_cast<const float*>(smem + SC_NOISE + (t & 3) * 1024)[(wave * 2 + nt) * 32 + l31];
            float o[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[nt][0][r] + (acc[nt][1][r] + acc[nt][2][r]);
                o[r] = v * osc[r];
                if (DOT) {
                    dsum[r] += (v * us) * dxv[nt][r];
                    if (PRE) o[r] *= dxv[nt][r] > 0.f ? kSqrt2 : 0.2f * kSqrt2;
                } else {
                    o[r] += nz + bia[r];
                    if (a.act == OODGAN_ACT_LRELU) o[r] = (o[r] > 0.f ? o[r] : 0.2f * o[r]) * kSqrt2;
                }
            }
            if (RGB) {
                // the lane's 16 channels of the three colour sums; the other 16 channels sit in lane ^ 32
                float c0s = 0.f, c1s = 0.f, c2s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    c0s += wr[0][r] * o[r];
                    c1s += wr[RGB ? 1 : 0][r] * o[r];
                    c2s += wr[RGB ? 2 : 0][r] * o[r];
                }
                c0s += __shfl_xor(c0s, 32, 64);
                c1s += __shfl_xor(c1s, 32, 64);
                c2s += __shfl_xor(c2s, 32, 64);
                if (ok && half == 0) {
                    float* rp = a.rgb_y + (long)b * 3 * H * W + (long)py * W + px;
                    rp[0] = c0s;
                    rp[(long)H * W] = c1s;
                    rp[2L * H * W] = c2s;
                }
            }
            if (STRIP_ABL & 1) {
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) q += o[r];
                if (q == 123456.75f) a.y[0] = q;
            } else if (ok && a.y_fform) {
                // F-form (oodgan_conv_args.y_fform): the lane's 16 channels are four float4 of the pixel's two 64-byte records
                // (channels 8i + 4*half .. + 3): 4 stores of 16 bytes instead of 16 of 4
                float* yf = a.y + ((((long)b * 2) * H + py) * W + px) * 16 + 4 * half;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    *reinterpret_cast<float4*>(yf + (long)(rr >> 1) * H * W * 16 + (rr & 1) * 8) =
                        make_float4(o[4 * rr], o[4 * rr + 1], o[4 * rr + 2], o[4 * rr + 3]);
            } else if (ok) {
                unsigned char* yr = ybt + (long)nt * a.out_pitch * 4;
                if (mfull) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(yr + moff[r]) = o[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((r & 3) + 8 * (r >> 2) + 4 * half < M) *reinterpret_cast<float*>(yr + moff[r]) = o[r];
                }
            }
        }
        if (DOT) {
            // sum over the 32 pixels of the row pair held by this half wave, then hand the 32 channel sums to LDS
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dsum[r] = half_sum_dpp(dsum[r]);
            }
            if (l31 == kHalfSumLane) {
                float* rp = red + (t & 1) * 128 + wave * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) rp[(r & 3) + 8 * (r >> 2) + 4 * half] = dsum[r];
            }
        }
    }
    if (DOT) {
        __syncthreads();
        if (wave == 0 && lane < 32 && lane < M) {
            const float* rp = red + ((n - 1) & 1) * 128;
            a.dot_part[((long)b * M + lane) * a.dot_nparts + (long)(t0 + n - 1) * p.tiles_x + tx] =
                rp[lane] + rp[32 + lane] + rp[64 + lane] + rp[96 + lane];
        }
    }
}

}  // namespace

namespace oodgan {

// true when the strip kernel can take this call (launch_s1v2 remains the general path)
bool s1_strip_eligible(const oodgan_conv_args& a) {
    return a.mode == OODGAN_CONV_S1 && a.x_sform && a.K > 16 && a.K <= 32 && a.M > 16 && a.M <= 32 && a.ys == nullptr &&
           a.y != nullptr && (a.act == OODGAN_ACT_NONE || a.act == OODGAN_ACT_LRELU) && a.in_scale == nullptr &&
           a.in_shift == nullptr && !(a.dotx && (a.noise || a.bias || a.act != OODGAN_ACT_NONE));
}

int launch_s1_strip(const oodgan_conv_args& a_in, const void* wpk16, const float* unscale, hipStream_t st) {
    StripConv p;
    p.a = a_in;
    oodgan_conv_args& a = p.a;
    if (a.out_pitch == 0) a.out_pitch = a.Win;
    p.out_plane = (long)a.Hin * a.out_pitch;
    p.xs = reinterpret_cast<const uint4*>(a.x);
    p.xd = sform_dims(a.K, a.Hin, a.Win);
    p.w_unscale = unscale;
    p.tiles_y = (a.Hin + 7) / 8;
    p.tiles_x = (a.Win + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    if (a.dotx) {
        OODGAN_REQUIRE(a.dot_part != nullptr, "conv3x3: dotx without dot_part");
        OODGAN_REQUIRE(a.dot_nparts == p.tiles_x * p.tiles_y, "conv3x3 f16s S1: dot_nparts %d != %d", a.dot_nparts,
                       p.tiles_x * p.tiles_y);
    }
    OODGAN_REQUIRE(!a.y_fform || (!a.dotx && a.M == 32 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0),
                   "conv3x3 strip: F-form output needs M == 32, no dotx and a 16-byte aligned y");
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            set_error("conv3x3 strip: cannot query the device");
            return OODGAN_E_LAUNCH;
        }
        num_cu = prop.multiProcessorCount;
    }
    // one workgroup per CU (143 KB of LDS): cut the strips into segments only when there are fewer strips than CUs
    const long strips = (long)a.B * p.tiles_x;
    int nseg = (int)((num_cu + strips - 1) / strips);
    if (nseg < 1) nseg = 1;
    int seg_tiles = (p.tiles_y + nseg - 1) / nseg;
    if (seg_tiles < 4) seg_tiles = p.tiles_y < 4 ? p.tiles_y : 4;
    p.seg_tiles = seg_tiles;
    p.nseg = (p.tiles_y + seg_tiles - 1) / seg_tiles;
    p.counted_wait = 1;
    const long nblk = strips * p.nseg;
    OODGAN_REQUIRE(nblk < (1L << 31), "conv3x3 strip: grid too large");
    OODGAN_REQUIRE((long)a.M * p.out_plane * 4 < (1L << 32) && (long)a.M * a.Hin * a.Win * 4 < (1L << 32), "conv3x3 strip: plane too large");
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_strip_kernel<true, false, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SC_SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_strip_kernel<true, false, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SC_SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_strip_kernel<false, false, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SC_SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_strip_kernel<false, true, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SC_SMEM), true);
    (void)once;
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
    if (a.dot_actgrad) {
        OODGAN_REQUIRE(a.dotx && !a.rgb_y, "conv3x3 strip: dot_actgrad needs dotx (and no fused ToRGB)");
        hipLaunchKernelGGL((conv_f16s_strip_kernel<true, false, true>), dim3((unsigned)nblk), dim3(256), SC_SMEM, st, p, w16);
    } else if (a.rgb_y) {
        OODGAN_REQUIRE(!a.dotx && a.rgb_w && a.rgb_s, "conv3x3 strip: the fused ToRGB needs rgb_w, rgb_s and no dotx");
        hipLaunchKernelGGL((conv_f16s_strip_kernel<false, true, false>), dim3((unsigned)nblk), dim3(256), SC_SMEM, st, p, w16);
    } else if (a.dotx) hipLaunchKernelGGL((conv_f16s_strip_kernel<true, false, false>), dim3((unsigned)nblk), dim3(256), SC_SMEM, st, p, w16);
    else hipLaunchKernelGGL((conv_f16s_strip_kernel<false, false, false>), dim3((unsigned)nblk), dim3(256), SC_SMEM, st, p, w16);
    return check_launch("conv3x3_f16s_strip");
}

}  // namespace oodgan
