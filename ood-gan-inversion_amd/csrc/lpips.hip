// LPIPS(net='alex') term of the W+ loss (round 6; north_star: "W+ Adam steps against LPIPS/L2").  Reference call site:
// src/losses/lpips_loss.py:13-34 — lpips.LPIPS(net='alex') on images normalised to [0,1] with normalize=True.  The `lpips` package and its
// weights are NOT part of the reference tree (un-vendored, unpinned): the arithmetic below restates the published algorithm
// (Zhang et al. 2018; lpips/lpips.py + pretrained_networks.py: scaling layer, torchvision AlexNet feature stack with taps after each of
// the five ReLUs, channel-unit normalisation with eps 1e-10, squared difference, non-negative 1x1 `lin` layers, spatial mean, sum over the
// taps) — PARITY UNPINNED, checked against oracle/lpips_cpu.py on seeded weights only.
//
// Everything is a stride-1 convolution here:
//   * conv1 (3 -> 64, 11x11, stride 4, pad 2) runs on the space-to-depth image: pad by 2, split into 4x4 pixel blocks ->
//     48 channels x (H/4+1)^2, and the 11x11 stride-4 kernel becomes a 3x3 stride-1 VALID kernel over those 48 channels (taps past
//     the eleventh are zero) — so its input gradient is a 3x3 "full" convolution followed by depth-to-space, not a 16-phase scatter;
//   * every input gradient of a stride-1 conv is the stride-1 conv with flipped, transposed weights and pad' = k - 1 - pad.
// One kernel, conv2d_s1_kernel<KS, CK>: implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 like the oracle;
// the term is opt-in and sized ~20 % of the generator's flops), 8 x 32 pixel tile x 64 output channels per 4-wave workgroup, operands
// staged through LDS with the next chunk's global loads in flight under the matrix loop.  Epilogue: + bias, ReLU (forward) or
// + add, x (mask > 0) (backward: the tap's own gradient joins the back-propagated one, then the ReLU below).
#include "conv_common.hpp"
#include <cstdint>

using namespace oodgan;

namespace {

struct Conv2dArgs {
    const float* x;
    const float* wpk;        // [K][KS*KS][Mp]
    const float* bias;       // (M) or NULL
    const float* add;        // (B,M,Hout,Wout) or NULL
    const float* mask;       // (B,M,Hout,Wout) or NULL: y = mask > 0 ? y : 0
    float* y;
    int B, K, M, Hin, Win, Hout, Wout, pad, relu, Mp, tiles_x, tiles_y, mblocks;
};

template <int KS, int CK>
__global__ __launch_bounds__(256) void conv2d_s1_kernel(const Conv2dArgs p) {
    constexpr int TR = 8, NT = 2, MT = 2, MB = 64;
    constexpr int IN_R = TR + KS - 1, IN_C = 32 + KS - 1, XT = IN_R * IN_C, XE = CK * XT;
    constexpr int XPT = (XE + 255) / 256;
    constexpr int TAPS = KS * KS, WROW = MB / 4, WE = CK * TAPS * WROW, WPT = (WE + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[XE + CK * TAPS * MB];
    float* lx = lds;
    float* lw = lds + XE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = p.tiles_x * p.tiles_y;
    const int tile = w % ntile;
    w /= ntile;
    const int b = w % p.B, mblk = w / p.B;
    const int r0 = (tile / p.tiles_x) * TR, c0 = (tile % p.tiles_x) * 32, m0 = mblk * MB;
    const long in_plane = (long)p.Hin * p.Win, out_plane = (long)p.Hout * p.Wout;
    const float* xb = p.x + (long)b * p.K * in_plane;

    int xoff[XPT], xch[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int e = tid + i * 256;
        const int c = e / XT, rem = e % XT;
        const int r = rem / IN_C, col = rem % IN_C;
        const int gy = r0 - p.pad + r, gx = c0 - p.pad + col;
        const bool ok = (e < XE) && gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win;
        xoff[i] = ok ? gy * p.Win + gx : -1;
        xch[i] = c;
    }
    float xr[XPT];
    float4 wr[WPT];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int k = k0 + xch[i];
            xr[i] = (xoff[i] >= 0 && k < p.K) ? xb[(long)k * in_plane + xoff[i]] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;
            const int row = e / WROW, q = e % WROW;        // row = c * TAPS + tap
            const int k = k0 + row / TAPS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < WE && k < p.K) v = *reinterpret_cast<const float4*>(p.wpk + ((long)k * TAPS + (row % TAPS)) * p.Mp + m0 + q * 4);
            wr[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            if (e < XE) lx[e] = xr[i];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;
            if (e < WE) reinterpret_cast<float4*>(lw)[e] = wr[i];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // lanes 0-31 feed input channel c, lanes 32-63 channel c+1 of the pair: one v_mfma_f32_32x32x2_f32 = two channels of one tap
    const float* lwh = lw + half * TAPS * MB + l31;
    const float* lxh = lx + half * XT + (wave * NT) * IN_C + l31;
    const int nchunk = (p.K + CK - 1) / CK;
    load_chunk(0);
    for (int t = 0; t < nchunk; ++t) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (t + 1 < nchunk) load_chunk((t + 1) * CK);
#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
            const float* xw = lxh + cp * 2 * XT;
            const float* ww = lwh + cp * 2 * TAPS * MB;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    float av[MT], bv[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) av[mt] = ww[(ky * KS + kx) * MB + mt * 32];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = xw[(nt + ky) * IN_C + kx];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
                }
        }
    }

    // epilogue: lane (l31, half) holds, per M-tile and row, channels (r & 3) + 8 (r >> 2) + 4 half of pixel column c0 + l31
    const int px = c0 + l31;
    float* yb = p.y + (long)b * p.M * out_plane;
    const float* ab = p.add ? p.add + (long)b * p.M * out_plane : nullptr;
    const float* kb = p.mask ? p.mask + (long)b * p.M * out_plane : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float bia[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.M - 1);
            bia[r] = p.bias ? p.bias[m] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int py = r0 + wave * NT + nt;
            if (py >= p.Hout || px >= p.Wout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= p.M) continue;
                const long idx = (long)m * out_plane + (long)py * p.Wout + px;
                float v = acc[mt][nt][r] + bia[r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (ab) v += ab[idx];
                if (kb) v = kb[idx] > 0.f ? v : 0.f;
                yb[idx] = v;
            }
        }
    }
}

// MaxPool2d(kernel 3, stride 2), no padding (torchvision AlexNet features[2], [5]).  idx (optional): position dy*3+dx of the window's FIRST maximum in
// row-major order — the element torch's max_pool2d routes the gradient to
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, long planes,
                                                             int H, int W, int Ho, int Wo) {
    const long total = planes * Ho * Wo;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % Wo), oy = (int)((e / Wo) % Ho);
        const long pl = e / ((long)Wo * Ho);
        const float* xp = x + pl * H * W + (long)(2 * oy) * W + 2 * ox;
        float m = xp[0];
        int arg = 0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float v = xp[dy * W + dx];
                if (v > m) { m = v; arg = dy * 3 + dx; }
            }
        y[e] = m;
        if (idx) idx[e] = (unsigned char)arg;
    }
}

// gx = (sum of gy over the windows whose first maximum is this element + add) * (x > 0) — x is a ReLU output: the mask is the ReLU backward of the
// layer that produced it.  idx != NULL: the argmax table of the forward (<= 4 byte loads per element); NULL: the windows are re-scanned (36 loads)
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ add,
                                                             const unsigned char* __restrict__ idx, float* __restrict__ gx, long planes, int H, int W,
                                                             int Ho, int Wo) {
    // block = 64 columns x 4 rows of one plane (grid: column blocks, row blocks, planes): no integer division per element — with the linear
    // grid-stride form the three divisions by 255 / 127 were a third of the kernel
    const int ix = blockIdx.x * 64 + (threadIdx.x & 63), iy = blockIdx.y * 4 + (threadIdx.x >> 6);
    for (long pl = blockIdx.z; pl < planes; pl += gridDim.z) {
        if (ix >= W || iy >= H) continue;
        const long e = (pl * H + iy) * W + ix;
        const float xv = x[e];
        float g = add ? add[e] : 0.f;
        const float* gp = gy + pl * Ho * Wo;
        if (idx) {
            // the <= 4 windows that hold (iy, ix): rows iy>>1 and, for even iy >= 2, (iy>>1) - 1; columns alike.  Unconditional loads from clamped
            // windows, all eight in flight together, masks at the point of use (a load inside the window loop is a dependent round trip per window)
            const unsigned char* ip = idx + pl * Ho * Wo;
            const int ya = min(iy >> 1, Ho - 1), yb = max((iy >> 1) - 1, 0), xa = min(ix >> 1, Wo - 1), xb = max((ix >> 1) - 1, 0);
            const bool vya = (iy >> 1) <= Ho - 1, vyb = !(iy & 1) && iy >= 2, vxa = (ix >> 1) <= Wo - 1, vxb = !(ix & 1) && ix >= 2;
            const int i0 = ip[(long)ya * Wo + xa], i1 = ip[(long)ya * Wo + xb], i2 = ip[(long)yb * Wo + xa], i3 = ip[(long)yb * Wo + xb];
            const float g0 = gp[(long)ya * Wo + xa], g1 = gp[(long)ya * Wo + xb], g2 = gp[(long)yb * Wo + xa], g3 = gp[(long)yb * Wo + xb];
            const int pya = (iy - 2 * ya) * 3, pyb = (iy - 2 * yb) * 3, pxa = ix - 2 * xa, pxb = ix - 2 * xb;
            g += (vya && vxa && i0 == pya + pxa) ? g0 : 0.f;
            g += (vya && vxb && i1 == pya + pxb) ? g1 : 0.f;
            g += (vyb && vxa && i2 == pyb + pxa) ? g2 : 0.f;
            g += (vyb && vxb && i3 == pyb + pxb) ? g3 : 0.f;
        } else {
            const float* xp = x + pl * H * W;
            const int oy0 = max((iy - 1) / 2, 0), oy1 = min(iy / 2, Ho - 1);      // windows with 2 oy <= iy <= 2 oy + 2
            const int ox0 = max((ix - 1) / 2, 0), ox1 = min(ix / 2, Wo - 1);
            for (int oy = oy0; oy <= oy1; ++oy)
                for (int ox = ox0; ox <= ox1; ++ox) {
                    if (2 * oy > iy || 2 * ox > ix) continue;
                    float m = -INFINITY;
                    int arg = -1;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const float v = xp[(long)(2 * oy + dy) * W + 2 * ox + dx];
                            if (v > m) { m = v; arg = dy * 3 + dx; }
                        }
                    if (arg == (iy - 2 * oy) * 3 + (ix - 2 * ox)) g += gp[(long)oy * Wo + ox];
                }
        }
        gx[e] = xv > 0.f ? g : 0.f;
    }
}

// image -> the conv1 operand: v = a x + b0 (min_max -> [-1,1]), ScalingLayer (v - shift_c) / scale_c, zero pad 2, 4x4 space-to-depth.
// out48 (B, 48, H/4 + 1, W/4 + 1), channel = c * 16 + dy * 4 + dx
__global__ __launch_bounds__(256) void lpips_prep_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int H, int W, float a,
                                                         float b0, float sh0, float sh1, float sh2, float is0, float is1, float is2) {
    const int Hs = H / 4 + 1, Ws = W / 4 + 1;
    const long total = (long)B * 48 * Hs * Ws;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int X = (int)(e % Ws), Y = (int)((e / Ws) % Hs);
        const int ch = (int)((e / ((long)Ws * Hs)) % 48);
        const int b = (int)(e / ((long)Ws * Hs * 48));
        const int c = ch >> 4, dy = (ch >> 2) & 3, dx = ch & 3;
        const int y = 4 * Y + dy - 2, x = 4 * X + dx - 2;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float sh = c == 0 ? sh0 : (c == 1 ? sh1 : sh2), is = c == 0 ? is0 : (c == 1 ? is1 : is2);
            v = (a * img[((long)(b * 3 + c) * H + y) * W + x] + b0 - sh) * is;
        }
        out[e] = v;
    }
}

// gimg += coef * a / scale_c * depth_to_space(g48)
__global__ __launch_bounds__(256) void lpips_img_grad_kernel(const float* __restrict__ g48, float* __restrict__ gimg, int B, int H, int W,
                                                             float k0, float k1, float k2) {
    const int Hs = H / 4 + 1, Ws = W / 4 + 1;
    const long total = (long)B * 3 * H * W;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % W), y = (int)((e / W) % H);
        const int c = (int)((e / ((long)W * H)) % 3);
        const int b = (int)(e / ((long)W * H * 3));
        const int ch = c * 16 + ((y + 2) & 3) * 4 + ((x + 2) & 3);
        const float g = g48[(((long)b * 48 + ch) * Hs + ((y + 2) >> 2)) * Ws + ((x + 2) >> 2)];
        gimg[e] += (c == 0 ? k0 : (c == 1 ? k1 : k2)) * g;
    }
}

// One tap of LPIPS.  Thread = pixel, three passes over the channels (NCHW: the lanes of a wave read consecutive pixels of one plane).
//   mode 0 (target): out = f / (sqrt(sum_c f^2) + 1e-10)                                   (lpips.normalize_tensor)
//   mode 1 (prediction): d = sum_c w_c (n0_c - n1_c)^2 -> part[b][block] (spatial sum), and
//          out = coef * d(d)/d(f0) = coef * ( r e_c - (sum_k e_k f0_k) r^2 / s * f0_c ),  e_c = 2 w_c (n0_c - n1_c), s = |f0|, r = 1/(s + eps)
//          (the second term is dropped where s == 0: torch's sqrt backward gives NaN there, a pixel with all C ReLU outputs at zero)
//   mode 2: mode 1 with out additionally multiplied by (f0 > 0) — the gradient w.r.t. the pre-activation of the LAST tap
// Block = 32 pixels x 8 channel groups (lane = pixel: a wave's loads of one plane are two 128-byte runs; the channel sums cross the groups through
// LDS): 8x the workgroups of a pixel-per-thread form — the 63² taps of a 1024² image are 3969 pixels x 256-384 channels per image, 128 workgroups
// of 256 threads on 256 CUs (322 us per tap, latency-bound) before, 1000 now.
constexpr int kHeadPix = 32, kHeadCg = 8, kHeadBlock = kHeadPix * kHeadCg;
__global__ __launch_bounds__(kHeadBlock) void lpips_head_kernel(const float* __restrict__ f0, const float* __restrict__ n1,
                                                                const float* __restrict__ w, float* __restrict__ out,
                                                                float* __restrict__ part, int C, long HW, float coef, int mode, int nblk) {
    __shared__ float red[2][kHeadCg][kHeadPix];
    const int b = blockIdx.y;
    const int pl = threadIdx.x & (kHeadPix - 1), cg = threadIdx.x / kHeadPix;
    const long pix = blockIdx.x * (long)kHeadPix + pl;
    const bool ok = pix < HW;
    const long p0 = ok ? pix : HW - 1;
    const float* fb = f0 + (long)b * C * HW + p0;
    float s2 = 0.f;
    for (int c = cg; c < C; c += kHeadCg) {
        const float v = fb[(long)c * HW];
        s2 += v * v;
    }
    red[0][cg][pl] = s2;
    __syncthreads();
    s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kHeadCg; ++k) s2 += red[0][k][pl];          // the same order in every group: one value per pixel
    const float s = sqrtf(s2), r = 1.f / (s + 1e-10f);
    float* ob = out + (long)b * C * HW + p0;
    if (mode == 0) {
        if (ok)
            for (int c = cg; c < C; c += kHeadCg) ob[(long)c * HW] = fb[(long)c * HW] * r;
        return;
    }
    const float* nb = n1 + (long)b * C * HW + p0;
    float d = 0.f, dotk = 0.f;
    for (int c = cg; c < C; c += kHeadCg) {
        const float f = fb[(long)c * HW], df = f * r - nb[(long)c * HW], wc = w[c];
        d += wc * df * df;
        dotk += 2.f * wc * df * f;
    }
    __syncthreads();                       // every group has read red[0]
    red[0][cg][pl] = d;
    red[1][cg][pl] = dotk;
    __syncthreads();
    d = dotk = 0.f;
#pragma unroll
    for (int k = 0; k < kHeadCg; ++k) { d += red[0][k][pl]; dotk += red[1][k][pl]; }
    const float q = s > 0.f ? dotk * r * r / s : 0.f;
    if (ok)
        for (int c = cg; c < C; c += kHeadCg) {
            const float f = fb[(long)c * HW], df = f * r - nb[(long)c * HW];
            const float gv = coef * (2.f * w[c] * df * r - q * f);
            ob[(long)c * HW] = (mode == 2 && !(f > 0.f)) ? 0.f : gv;      // mode 2: the deepest tap — nothing joins it, its own ReLU mask here
        }
    // deterministic block sum of d over the block's pixels (group 0 holds the per-pixel totals)
    if (cg == 0) {
        float v = ok ? d : 0.f;
#pragma unroll
        for (int o = kHeadPix / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (pl == 0) part[(long)b * nblk + blockIdx.x] = v;
    }
}

// y = (y + add) * (mask > 0), in place: the tap's gradient joins the back-propagated one, then the ReLU below — behind an input-gradient conv that ran
// on the split-f16 kernels of oodgan_conv3x3_f16s (their epilogue has no such term)
__global__ __launch_bounds__(256) void add_mask_kernel(float* __restrict__ y, const float* __restrict__ add, const float* __restrict__ mask, long n4) {
    const long stride = (long)gridDim.x * 256;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4*>(y)[i];
        const float4 a = reinterpret_cast<const float4*>(add)[i], m = reinterpret_cast<const float4*>(mask)[i];
        v.x = m.x > 0.f ? v.x + a.x : 0.f;
        v.y = m.y > 0.f ? v.y + a.y : 0.f;
        v.z = m.z > 0.f ? v.z + a.z : 0.f;
        v.w = m.w > 0.f ? v.w + a.w : 0.f;
        reinterpret_cast<float4*>(y)[i] = v;
    }
}
__global__ __launch_bounds__(256) void add_mask_tail_kernel(float* __restrict__ y, const float* __restrict__ add, const float* __restrict__ mask, long n0, long n) {
    const long i = n0 + blockIdx.x * 256L + threadIdx.x;
    if (i < n) y[i] = mask[i] > 0.f ? y[i] + add[i] : 0.f;
}

struct FinishArgs {
    const float* part[5];
    int nparts[5];
    float inv_hw[5];
    int ntaps;
};
// lpips[b] = sum_taps (sum_blocks part) / HW_tap  -> row min(row_dev[0], nrows-1) of table (nrows, B) (row_dev NULL: row 0)
__global__ __launch_bounds__(64) void lpips_finish_kernel(const FinishArgs a, float* __restrict__ table, const int* __restrict__ row_dev,
                                                          int nrows) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float tot = 0.f;
    for (int t = 0; t < a.ntaps; ++t) {
        float s = 0.f;
        for (int j = lane; j < a.nparts[t]; j += 64) s += a.part[t][(long)b * a.nparts[t] + j];
        tot += wave_sum(s) * a.inv_hw[t];
    }
    const long row = row_dev ? (long)min(max(row_dev[0], 0), nrows - 1) * gridDim.x : 0;
    if (lane == 0) table[row + b] = tot;
}

}  // namespace

extern "C" int oodgan_conv2d_s1(const float* x, const float* wpk, const float* bias, const float* add, const float* mask, float* y, int B,
                                int K, int M, int Hin, int Win, int ks, int pad, int relu, void* stream) {
    OODGAN_REQUIRE(x && wpk && y && B > 0 && K > 0 && M > 0 && Hin > 0 && Win > 0, "conv2d_s1: bad args");
    OODGAN_REQUIRE((ks == 3 || ks == 5) && pad >= 0 && pad < ks, "conv2d_s1: kernel size %d / pad %d not supported (3 or 5, 0 <= pad < ks)", ks, pad);
    Conv2dArgs p;
    p.x = x; p.wpk = wpk; p.bias = bias; p.add = add; p.mask = mask; p.y = y;
    p.B = B; p.K = K; p.M = M; p.Hin = Hin; p.Win = Win; p.pad = pad; p.relu = relu;
    p.Hout = Hin + 2 * pad - ks + 1;
    p.Wout = Win + 2 * pad - ks + 1;
    OODGAN_REQUIRE(p.Hout > 0 && p.Wout > 0, "conv2d_s1: empty output");
    OODGAN_REQUIRE((long)Hin * Win < (1L << 31) && (long)M * p.Hout * p.Wout < (1L << 40), "conv2d_s1: plane too large");
    p.Mp = (M + 63) / 64 * 64;
    p.mblocks = (M + 63) / 64;
    p.tiles_y = (p.Hout + 7) / 8;
    p.tiles_x = (p.Wout + 31) / 32;
    const long total = (long)p.tiles_x * p.tiles_y * B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv2d_s1: grid too large");
    if (ks == 3) hipLaunchKernelGGL((conv2d_s1_kernel<3, 8>), dim3((unsigned)total), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL((conv2d_s1_kernel<5, 4>), dim3((unsigned)total), dim3(256), 0, as_stream(stream), p);
    return check_launch("conv2d_s1");
}

extern "C" int oodgan_add_mask(float* y, const float* add, const float* mask, long n, void* stream) {
    OODGAN_REQUIRE(y && add && mask && n > 0, "add_mask: bad args");
    OODGAN_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(add) | reinterpret_cast<uintptr_t>(mask)) & 15) == 0, "add_mask: 16-byte aligned tensors");
    const long n4 = n / 4;
    if (n4 > 0) hipLaunchKernelGGL(add_mask_kernel, dim3(stream_grid(n4, 256)), dim3(256), 0, as_stream(stream), y, add, mask, n4);
    if (n4 * 4 < n) hipLaunchKernelGGL(add_mask_tail_kernel, dim3(1), dim3(256), 0, as_stream(stream), y, add, mask, n4 * 4, n);
    return check_launch("add_mask");
}

extern "C" int oodgan_maxpool3s2_fwd(const float* x, float* y, unsigned char* idx, long planes, int H, int W, void* stream) {
    OODGAN_REQUIRE(x && y && planes > 0 && H >= 3 && W >= 3, "maxpool3s2_fwd: bad args");
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(stream_grid(planes * Ho * Wo, 256)), dim3(256), 0, as_stream(stream), x, y, idx, planes, H, W, Ho, Wo);
    return check_launch("maxpool3s2_fwd");
}

extern "C" int oodgan_maxpool3s2_bwd(const float* x, const float* gy, const float* add, const unsigned char* idx, float* gx, long planes, int H, int W,
                                     void* stream) {
    OODGAN_REQUIRE(x && gy && gx && planes > 0 && H >= 3 && W >= 3, "maxpool3s2_bwd: bad args");
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const unsigned gz = (unsigned)(planes < 32768 ? planes : 32768);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3((W + 63) / 64, (H + 3) / 4, gz), dim3(256), 0, as_stream(stream), x, gy, add, idx, gx, planes, H, W, Ho, Wo);
    return check_launch("maxpool3s2_bwd");
}

extern "C" int oodgan_lpips_prep(const float* img, float* out48, int B, int H, int W, float a, float b0, const float* shift3,
                                 const float* scale3, void* stream) {
    OODGAN_REQUIRE(img && out48 && shift3 && scale3 && B > 0 && H >= 16 && W >= 16 && H % 4 == 0 && W % 4 == 0, "lpips_prep: bad args (H, W multiples of 4, >= 16)");
    const long total = (long)B * 48 * (H / 4 + 1) * (W / 4 + 1);
    // host arrays are read HERE: a recorded launch (oodgan_plan_*) keeps the values, not the pointers
    const float sh0 = shift3[0], sh1 = shift3[1], sh2 = shift3[2], is0 = 1.f / scale3[0], is1 = 1.f / scale3[1], is2 = 1.f / scale3[2];
    hipLaunchKernelGGL(lpips_prep_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), img, out48, B, H, W, a, b0, sh0, sh1, sh2,
                       is0, is1, is2);
    return check_launch("lpips_prep");
}

extern "C" int oodgan_lpips_img_grad(const float* g48, float* gimg, int B, int H, int W, float a, float coef, const float* scale3, void* stream) {
    OODGAN_REQUIRE(g48 && gimg && scale3 && B > 0 && H % 4 == 0 && W % 4 == 0, "lpips_img_grad: bad args");
    const float k0 = coef * a / scale3[0], k1 = coef * a / scale3[1], k2 = coef * a / scale3[2];
    hipLaunchKernelGGL(lpips_img_grad_kernel, dim3(stream_grid((long)B * 3 * H * W, 256)), dim3(256), 0, as_stream(stream), g48, gimg, B, H, W, k0, k1,
                       k2);
    return check_launch("lpips_img_grad");
}

extern "C" int oodgan_lpips_head_nparts(long HW) { return (int)((HW + kHeadPix - 1) / kHeadPix); }

extern "C" int oodgan_lpips_head(const float* f0, const float* n1, const float* w, float* out, float* part, int B, int C, long HW, float coef,
                                 int mode, void* stream) {
    OODGAN_REQUIRE(f0 && out && B > 0 && B < 65536 && C > 0 && HW > 0 && (mode == 0 || ((mode == 1 || mode == 2) && n1 && w && part)), "lpips_head: bad args");
    const int nblk = oodgan_lpips_head_nparts(HW);
    hipLaunchKernelGGL(lpips_head_kernel, dim3(nblk, B), dim3(kHeadBlock), 0, as_stream(stream), f0, n1, w, out, part, C, HW, coef, mode, nblk);
    return check_launch("lpips_head");
}

extern "C" int oodgan_lpips_finish(const float* const* parts, const int* nparts, const long* hw, int ntaps, float* table, const int* row_dev,
                                   int nrows, int B, void* stream) {
    OODGAN_REQUIRE(parts && nparts && hw && ntaps > 0 && ntaps <= 5 && table && B > 0 && nrows > 0, "lpips_finish: bad args");
    FinishArgs a;
    a.ntaps = ntaps;
    for (int t = 0; t < 5; ++t) {
        a.part[t] = t < ntaps ? parts[t] : nullptr;
        a.nparts[t] = t < ntaps ? nparts[t] : 0;
        a.inv_hw[t] = t < ntaps ? 1.f / (float)hw[t] : 0.f;
    }
    hipLaunchKernelGGL(lpips_finish_kernel, dim3(B), dim3(64), 0, as_stream(stream), a, table, row_dev, nrows);
    return check_launch("lpips_finish");
}
