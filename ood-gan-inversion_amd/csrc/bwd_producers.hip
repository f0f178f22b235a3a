// Backward "producers": the gradient of the StyledConv tail (bias + noise + leaky-ReLU*sqrt2, merged with the ToRGB
// branch — the same arithmetic as act_bwd_fused_kernel in elementwise.hip) written DIRECTLY in the layout the next
// matrix kernel consumes, so the fp32 pre-activation gradient never goes to HBM:
//   * act_bwd_sform_kernel      -> S-form of g_pre*d*scale            (input of the plain 3x3 input-gradient conv)
//   * act_bwd_blurT_sp_kernel   -> phase-split S-form of blur^T(g_pre)*d*scale  (input of the stride-2 conv that is the
//                                  input gradient of the up-sampling ModulatedConv2d)
// The power-of-two range scale of the split-f16 format (mul2 = {2^-e, 2^e}) cannot be derived from this pass's own
// maximum without a second pass, so the callers hand in the scale measured on the PREVIOUS optimisation step; the
// kernels still reduce max|g_pre| and oodgan_absmax_scale_check verifies that the value actually stayed inside the
// window in which the f16 split is exact to fp32 (and flags the step otherwise) before publishing the next scale.
// Reference semantics: autograd of FusedLeakyReLU / NoiseInjection / ToRGB (src/ops/StyleGAN/model.py:283-292,343-372,
// src/ops/op/fused_act.py:25-58) and of Blur(pad=(1,1)) (src/ops/op/upfirdn2d.py:115-120).
#include "common.hpp"
#include "sform.hpp"
#include <cstdint>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

namespace {

struct SPDims { int KC, Hq, Wq; long plane; };

__host__ __device__ inline SPDims sp_dims(int C, int H, int W) {    // must match conv_f16s_v2.hip
    SPDims d;
    d.KC = (C + 15) / 16;
    d.Hq = (H + 7) / 8 * 8 + 2;
    d.Wq = (W + 31) / 32 * 32 + 2;
    d.plane = (long)d.Hq * d.Wq * 4;
    return d;
}

struct ActArgs {
    const float* g_feat;     // (B,C,H,W) or null
    const float* out;        // (B,C,H,W) post-activation
    const float* noise;      // (noise_batch,H,W) or null
    const float* noise_w;
    const float* bias;       // (C) or null
    const float* g_rgb;      // (B,3,H,W) or null
    const float* w_rgb;      // (3,C)
    const float* s_rgb;      // (B,*) stride s_rgb_stride
    const float* dscale;     // (B,*) stride dscale_stride: demodulation of this layer
    const float* mul2;       // device {unscale, scale}: the range scale USED by this pass
    float* part_r;           // (B,C,nparts)
    float* part_t;           // (B,C,nparts) or null
    float* part_max;         // (B*KC*nparts)
    int noise_batch, s_rgb_stride, dscale_stride;
    float rgb_scale;
    int B, C, H, W, nparts;
};

constexpr float kInvPos = 1.f / kSqrt2, kInvNeg = 1.f / (0.2f * kSqrt2);

// per-block channel constants in LDS: [0] w0 [1] w1 [2] w2 (already x rgb_scale) [3] s_rgb [4] bias [5] d*scale [6] |d|
__device__ __forceinline__ void load_consts(const ActArgs& a, int b, int kc, float (*cst)[16]) {
    const int t = threadIdx.x;
    if (t < 112) {
        const int which = t >> 4, j = t & 15, c = kc * 16 + j;
        float v = 0.f;
        if (c < a.C) {
            if (which < 3) v = a.g_rgb ? a.w_rgb[which * a.C + c] * a.rgb_scale : 0.f;
            else if (which == 3) v = a.g_rgb ? a.s_rgb[(long)b * a.s_rgb_stride + c] : 0.f;
            else if (which == 4) v = a.bias ? a.bias[c] : 0.f;
            else if (which == 5) v = (a.dscale ? a.dscale[(long)b * a.dscale_stride + c] : 1.f) * (a.mul2 ? a.mul2[1] : 1.f);
            else v = a.dscale ? fabsf(a.dscale[(long)b * a.dscale_stride + c]) : 1.f;
        }
        cst[which][j] = v;
    }
}

constexpr int kP1Chunk = 512;       // pixels of one (b, 16-channel block) per workgroup
constexpr int kP1Pitch = kP1Chunk + 4;

// grid (nparts, B*KC).  Phase A: thread = (channel tid>>4, 16 threads per channel) -> float4 loads along the plane,
// per-channel reductions stay inside 16 lanes; the scaled gradient goes to LDS [channel][pixel].  Phase C: thread =
// pixel, gathers its 16 channels, splits hi/lo and writes the 64-byte record (consecutive lanes, consecutive records).
__global__ __launch_bounds__(256) void act_bwd_sform_kernel(const ActArgs a, uint4* __restrict__ ys, const SDims yd) {
    __shared__ __attribute__((aligned(16))) float lst[16 * kP1Pitch];
    __shared__ float cst[7][16];
    __shared__ float redm[4];
    const int KC = yd.KC;
    const int bk = blockIdx.y, b = bk / KC, kc = bk % KC;
    const int tid = threadIdx.x;
    const long HW = (long)a.H * a.W;
    load_consts(a, b, kc, cst);
    __syncthreads();
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const float* np = a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * HW : nullptr;
    const float* gr = a.g_rgb ? a.g_rgb + (long)b * 3 * HW : nullptr;
    const long p0 = (long)blockIdx.x * kP1Chunk;
    float amax = 0.f;
    {
        const int ch = tid >> 4, q = tid & 15, c = kc * 16 + ch;
        const float w0 = cst[0][ch], w1 = cst[1][ch], w2 = cst[2][ch], sr = cst[3][ch], bv = cst[4][ch], ds = cst[5][ch];
        const long cbase = ((long)b * a.C + c) * HW;
        float acc_r = 0.f, acc_t = 0.f;
#pragma unroll
        for (int k = 0; k < kP1Chunk / 64; ++k) {
            const int lp = 4 * (q + 16 * k);
            const long p = p0 + lp;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < a.C && p < HW) {
                const float4 o4 = *reinterpret_cast<const float4*>(a.out + cbase + p);
                float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), nz = g4, r0 = g4, r1 = g4, r2 = g4;
                if (a.g_feat) g4 = *reinterpret_cast<const float4*>(a.g_feat + cbase + p);
                if (np) nz = *reinterpret_cast<const float4*>(np + p);
                if (gr) {
                    r0 = *reinterpret_cast<const float4*>(gr + p);
                    r1 = *reinterpret_cast<const float4*>(gr + HW + p);
                    r2 = *reinterpret_cast<const float4*>(gr + 2 * HW + p);
                }
                const float ov[4] = {o4.x, o4.y, o4.z, o4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w}, nzv[4] = {nz.x, nz.y, nz.z, nz.w};
                const float r0v[4] = {r0.x, r0.y, r0.z, r0.w}, r1v[4] = {r1.x, r1.y, r1.z, r1.w}, r2v[4] = {r2.x, r2.y, r2.z, r2.w};
                float gp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float o = ov[j];
                    const float t = w0 * r0v[j] + w1 * r1v[j] + w2 * r2v[j];
                    const float g = gv[j] + sr * t;
                    gp[j] = g * (o > 0.f ? kSqrt2 : 0.2f * kSqrt2);
                    const float ycv = (o > 0.f ? o * kInvPos : o * kInvNeg) - nw * nzv[j] - bv;
                    acc_r += gp[j] * ycv;
                    acc_t += o * t;
                    amax = fmaxf(amax, fabsf(gp[j]));
                }
                v = make_float4(gp[0] * ds, gp[1] * ds, gp[2] * ds, gp[3] * ds);
            }
            *reinterpret_cast<float4*>(lst + ch * kP1Pitch + lp) = v;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            acc_r += __shfl_xor(acc_r, o, 64);
            acc_t += __shfl_xor(acc_t, o, 64);
        }
        if (q == 0 && c < a.C) {
            const long o = ((long)b * a.C + c) * a.nparts + blockIdx.x;
            a.part_r[o] = acc_r;
            if (a.part_t) a.part_t[o] = acc_t;
        }
    }
    amax *= cst[6][tid >> 4];        // the value written is g_pre*d*scale: the range scale must cover the demodulation factor
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((tid & 63) == 0) redm[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) a.part_max[(long)bk * a.nparts + blockIdx.x] = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
#pragma unroll
    for (int rep = 0; rep < kP1Chunk / 256; ++rep) {
        const int lp = tid + rep * 256;
        const long p = p0 + lp;
        if (p >= HW) continue;
        unsigned hp[8], lq[8];
#pragma unroll
        for (int cp = 0; cp < 8; ++cp) {
            const float v0 = lst[(2 * cp) * kP1Pitch + lp], v1 = lst[(2 * cp + 1) * kP1Pitch + lp];
            split_pair(v0, v1, hp[cp], lq[cp]);
        }
        const int y = (int)(p / a.W), x = (int)(p % a.W);
        uint4* rec = ys + sform_unit(yd, b, kc, y, x, 0);
        rec[0] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
        rec[1] = make_uint4(hp[4], hp[5], hp[6], hp[7]);
        rec[2] = make_uint4(lq[0], lq[1], lq[2], lq[3]);
        rec[3] = make_uint4(lq[4], lq[5], lq[6], lq[7]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused act-backward + blur^T + phase split (see blurT_sp_kernel in conv_f16s_v2.hip for the tiling): block = one
// (b, 16-channel block) x a tile of 4x32 (i,j) positions = g2 rows 2*i0..2*i0+7, cols 2*j0..2*j0+63.  H,W = size of
// the up-conv's INPUT; all tensors of ActArgs are at (2H)x(2W).
constexpr int BT_R = 11, BT_C = 72;
// LDS pitches (floats): row 74, channel plane 816 — with the FIR's lane mapping (4 channels x 8 rows x 2 half rows per wave)
// every ds_read_b64 of a half wave covers the 64 banks once (72 / 792 was a 4-way conflict).  Exchange buffer:
// [16 ch x 521][4 phases x 130][4 rows x 32, column ^ (row >= 2 ? 8 : 0)] — conflict-free for the FIR's scalar writes and the
// pixel-major reads.
constexpr int BT_P = 74, BT_Q = 816, BX_CH = 521, BX_PH = 130;

// element k (0..12) of thread i (0..15 within its channel) of the 11-row x 18-float4 input tile: the thread walks down
// column group i (k = row), then rows 0..10 of the two halo column groups 16 and 17 go to threads 0..10.  Row offsets are
// compile-time and the column is fixed per thread, so the sweep needs no per-element division or bounds arithmetic.
__device__ __forceinline__ bool tile_elem(int k, int i, int& r, int& c4) {
    if (k < BT_R) { r = k; c4 = i; return true; }
    r = i; c4 = 16 + (k - BT_R);
    return i < BT_R;
}

template <bool RGB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void act_bwd_blurT_sp_kernel(const ActArgs a, const float* __restrict__ kern,
                                                               uint4* __restrict__ outp, int H, int W, SPDims sp, int tiles_x,
                                                               int tiles_y) {
    __shared__ __attribute__((aligned(16))) float lin[16 * BT_Q];
    __shared__ float kf[16];
    __shared__ float ksep[9];
    __shared__ float cst[7][16];
    __shared__ float redm[4];
    const int tid = threadIdx.x;
    // contiguous chunk of the tile list per XCD: a tile's halo rows/columns are its neighbours' interiors, which the
    // same L2 then already holds
    int w;
    {
        const int total = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = w % tiles_x; w /= tiles_x;
    const int ty = w % tiles_y; w /= tiles_y;
    const int kc = w % sp.KC;
    const int b = w / sp.KC;
    const int tile = ty * tiles_x + tx;
    const int i0 = ty * 4, j0 = tx * 32;
    const int Hg = 2 * H, Wg = 2 * W;
    const long HW = (long)Hg * Wg;
    if (tid < 16) kf[tid] = kern[tid];
    load_consts(a, b, kc, cst);
    __syncthreads();
    if (tid == 0) {
        // rank-1 test: kern[a][b] == u[a]*v[b] with u = column 0 / kern[0][0], v = row 0 (exact for outer products)
        bool ok = kf[0] != 0.f;
        for (int i = 0; i < 4 && ok; ++i) {
            ksep[i] = kf[i * 4] / kf[0];
            ksep[4 + i] = kf[i];
        }
        for (int i = 0; i < 16 && ok; ++i) ok = fabsf(ksep[i >> 2] * ksep[4 + (i & 3)] - kf[i]) <= 1e-7f * fabsf(kf[0]);
        ksep[8] = ok ? 1.f : 0.f;
    }
    __syncthreads();
    const bool sep_ok = ksep[8] != 0.f;
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const float* np = a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * HW : nullptr;
    const float* gr = (RGB && a.g_rgb) ? a.g_rgb + (long)b * 3 * HW : nullptr;
    // ---- A: thread -> fixed channel ch = tid>>4, 16 threads sweep its 11 x 18 float4 tile
    const int gy0 = 2 * i0 - 2, gx0 = 2 * j0 - 4;
    float amax = 0.f;
    {
        const int ch = tid >> 4, c = kc * 16 + ch;
        const float w0 = cst[0][ch], w1 = cst[1][ch], w2 = cst[2][ch], sr = cst[3][ch], bv = cst[4][ch];
        const long cbase = ((long)b * a.C + c) * HW;
        float acc_r = 0.f, acc_t = 0.f;
        // loads are issued in two batches of 7 / 6 tile elements before anything depends on them (a load per iteration
        // would serialise the fill into 13 HBM latencies)
        constexpr int NE = BT_R + 2;                              // 13 (tile_elem)
        constexpr int NB = RGB ? 4 : 7;                           // batch: NB elements x (3 or 6) float4 loads in flight
#pragma unroll
        for (int k0 = 0; k0 < NE; k0 += NB) {
            float4 o4[NB], g4[NB], nz4[NB], r04[RGB ? NB : 1], r14[RGB ? NB : 1], r24[RGB ? NB : 1];
            bool okv[NB];
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) {
                int r, c4;
                const bool act = tile_elem(k0 + kk, tid & 15, r, c4);
                const int gy = gy0 + r, gx = gx0 + 4 * c4;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                o4[kk] = g4[kk] = nz4[kk] = z4;
                if (RGB) r04[kk] = r14[kk] = r24[kk] = z4;
                okv[kk] = k0 + kk < NE && act && c < a.C && gy >= 0 && gy < Hg && gx >= 0 && gx + 3 < Wg;
                if (okv[kk]) {
                    const long p = (long)gy * Wg + gx;
                    o4[kk] = *reinterpret_cast<const float4*>(a.out + cbase + p);
                    if (a.g_feat) g4[kk] = *reinterpret_cast<const float4*>(a.g_feat + cbase + p);
                    if (np) nz4[kk] = *reinterpret_cast<const float4*>(np + p);
                    if (RGB && gr) {
                        r04[kk] = *reinterpret_cast<const float4*>(gr + p);
                        r14[kk] = *reinterpret_cast<const float4*>(gr + HW + p);
                        r24[kk] = *reinterpret_cast<const float4*>(gr + 2 * HW + p);
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) {
                int r, c4;
                if (k0 + kk >= NE || !tile_elem(k0 + kk, tid & 15, r, c4)) continue;
                const int gy = gy0 + r, gx = gx0 + 4 * c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (okv[kk]) {
                    const float ov[4] = {o4[kk].x, o4[kk].y, o4[kk].z, o4[kk].w}, gv[4] = {g4[kk].x, g4[kk].y, g4[kk].z, g4[kk].w};
                    const float nzv[4] = {nz4[kk].x, nz4[kk].y, nz4[kk].z, nz4[kk].w};
                    constexpr int ri = RGB ? 1 : 0;
                    const float4 q0 = RGB ? r04[kk * ri] : make_float4(0.f, 0.f, 0.f, 0.f), q1 = RGB ? r14[kk * ri] : q0, q2 = RGB ? r24[kk * ri] : q0;
                    const float r0v[4] = {q0.x, q0.y, q0.z, q0.w}, r1v[4] = {q1.x, q1.y, q1.z, q1.w}, r2v[4] = {q2.x, q2.y, q2.z, q2.w};
                    // every g pixel is reduced by exactly one block: the one whose 8x64 interior contains it
                    const bool own = gy >= 2 * i0 && gy < 2 * i0 + 8 && gx >= 2 * j0 && gx < 2 * j0 + 64;
                    float gp[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float o = ov[j];
                        float t = 0.f, g = gv[j];
                        if (RGB) {          // without a ToRGB branch the three products are not even issued
                            t = w0 * r0v[j] + w1 * r1v[j] + w2 * r2v[j];
                            g += sr * t;
                        }
                        gp[j] = g * (o > 0.f ? kSqrt2 : 0.2f * kSqrt2);
                        if (own) {
                            const float ycv = (o > 0.f ? o * kInvPos : o * kInvNeg) - nw * nzv[j] - bv;
                            acc_r += gp[j] * ycv;
                            if (RGB) acc_t += o * t;
                            amax = fmaxf(amax, fabsf(gp[j]));
                        }
                    }
                    v = make_float4(gp[0], gp[1], gp[2], gp[3]);
                }
                float2* dst = reinterpret_cast<float2*>(lin + ch * BT_Q + r * BT_P + 4 * c4);
                dst[0] = make_float2(v.x, v.y);
                dst[1] = make_float2(v.z, v.w);
            }
        }
        // reduce over the 16 threads of the channel
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            acc_r += __shfl_xor(acc_r, o, 64);
            acc_t += __shfl_xor(acc_t, o, 64);
        }
        if ((tid & 15) == 0 && c < a.C) {
            const long o = ((long)b * a.C + c) * a.nparts + tile;
            a.part_r[o] = acc_r;
            if (a.part_t) a.part_t[o] = acc_t;
        }
    }
    amax *= cst[6][tid >> 4];        // the value written is g_pre*d*scale: the range scale must cover the demodulation factor
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((tid & 63) == 0) redm[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) a.part_max[((long)b * sp.KC + kc) * a.nparts + tile] = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    // ---- B: g2[Y,X] = sum_{a,b} kflip[a][b] * g[Y+a-2, X+b-2]; a rank-1 kernel (the [1,3,3,1] blur is one) is applied
    // as a vertical pass over the 35-wide window followed by a horizontal pass: 268 instead of 512 FMAs per thread
    const int ch = tid >> 4, tq = tid & 15;
    const int yrow = tq >> 1, xh = tq & 1;
    float o[32];
    if (sep_ok) {
        float kv[4], kh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { kv[i] = ksep[3 - i]; kh[i] = ksep[4 + 3 - i]; }
        // two halves of 16 outputs, a 20-wide vertical pass each (keeps the live set at 32 + 20 registers)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float tmp[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) tmp[j] = 0.f;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const float2* row = reinterpret_cast<const float2*>(lin + ch * BT_Q + (yrow + aa) * BT_P + 32 * xh + 16 * hf + 2);
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const float2 v = row[j];
                    tmp[2 * j] += kv[aa] * v.x;
                    tmp[2 * j + 1] += kv[aa] * v.y;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                o[16 * hf + j] = kh[0] * tmp[j] + kh[1] * tmp[j + 1] + kh[2] * tmp[j + 2] + kh[3] * tmp[j + 3];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) o[j] = 0.f;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            const float* row = lin + ch * BT_Q + (yrow + aa) * BT_P + 32 * xh + 2;
            float win[35];
#pragma unroll
            for (int j = 0; j < 35; ++j) win[j] = row[j];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float kv = kf[(3 - aa) * 4 + (3 - bb)];
#pragma unroll
                for (int j = 0; j < 32; ++j) o[j] += kv * win[j + bb];
            }
        }
    }
    const float sc_ = cst[5][ch];
    __syncthreads();
    float* lst = lin;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int Xl = 32 * xh + j;
        const int ph = (yrow & 1) * 2 + (Xl & 1);
        lst[ch * BX_CH + ph * BX_PH + (yrow >> 1) * 32 + ((Xl >> 1) ^ ((yrow & 4) << 1))] = o[j] * sc_;
    }
    __syncthreads();
    // ---- C
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int pos = tid + rep * 256;
        const int ph = pos >> 7, il = (pos >> 5) & 3, jl = pos & 31;
        const int i = i0 + il, j = j0 + jl;
        if (i > H || j > W) continue;
        half8 h0, h1, l0, l1;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            float v = lst[cc * BX_CH + ph * BX_PH + il * 32 + (jl ^ ((il & 2) << 2))];
            const int Y = 2 * i + (ph >> 1), X = 2 * j + (ph & 1);
            if (Y > 2 * H || X > 2 * W) v = 0.f;
            const _Float16 hh = (_Float16)v;
            const _Float16 ll = (_Float16)(v - (float)hh);
            if (cc < 8) { h0[cc] = hh; l0[cc] = ll; } else { h1[cc - 8] = hh; l1[cc - 8] = ll; }
        }
        half8* rec = reinterpret_cast<half8*>(outp + ((((long)b * sp.KC + kc) * 4 + ph) * sp.plane + ((long)i * sp.Wq + j) * 4));
        rec[0] = h0; rec[1] = h1; rec[2] = l0; rec[3] = l1;
    }
}

// max over the per-block partial maxima -> next range scale; verifies the scale that was USED.
// state[0..1] = {unscale, scale} (in: used by the pass that produced `part`, out: for the next step);
// flag[0] |= 1 when max*scale_used left [2^-8, 2^15] (the window in which hi+lo is exact to fp32 and cannot overflow
// after the demodulation factor), |= 2 when a non-finite value was seen.
__global__ __launch_bounds__(1024) void absmax_scale_check_kernel(const float* __restrict__ part, long n, float* __restrict__ state,
                                                                  int* __restrict__ flag) {
    __shared__ float red[16];
    float m = 0.f;
    bool bad = false;
    // 4 independent loads in flight per thread: the list has up to ~130k entries and one block reads it
    for (long i = threadIdx.x; i < n; i += 4096) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + 1024 * u < n ? part[i + 1024 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!isfinite(v[u])) bad = true;
            m = fmaxf(m, fabsf(v[u]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 2);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
        const float used = state[1];
        if (m > 0.f && isfinite(m)) {
            const float scaled = m * used;
            if (!(scaled >= 0.00390625f && scaled < 32768.f)) atomicOr(flag, 1);
        }
        int e = 0;
        if (m > 0.f && isfinite(m)) e = 9 - (int)floorf(log2f(m));
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        state[0] = ldexpf(1.f, -e);
        state[1] = ldexpf(1.f, e);
    }
}

int fill_args(ActArgs& a, const float* g_feat, const float* out, const float* noise, int noise_batch, const float* noise_w,
              const float* bias, const float* g_rgb, const float* w_rgb, const float* s_rgb, int s_rgb_stride, float rgb_scale,
              const float* dscale, int dscale_stride, const float* mul2, float* part_r, float* part_t, float* part_max, int B,
              int C, int H, int W) {
    OODGAN_REQUIRE(out && part_r && part_max && mul2 && B > 0 && C > 0 && H > 0 && W > 0, "act_bwd producer: bad args");
    OODGAN_REQUIRE(!g_rgb || (w_rgb && s_rgb && part_t), "act_bwd producer: rgb branch needs w_rgb, s_rgb and part_t");
    OODGAN_REQUIRE(noise == nullptr || noise_batch == 1 || noise_batch == B, "act_bwd producer: noise_batch");
    OODGAN_REQUIRE((W % 4) == 0, "act_bwd producer: W must be a multiple of 4");
    a.g_feat = g_feat; a.out = out; a.noise = noise; a.noise_w = noise_w; a.bias = bias;
    a.g_rgb = g_rgb; a.w_rgb = w_rgb; a.s_rgb = s_rgb; a.dscale = dscale; a.mul2 = mul2;
    a.part_r = part_r; a.part_t = g_rgb ? part_t : nullptr; a.part_max = part_max;
    a.noise_batch = noise_batch; a.s_rgb_stride = s_rgb_stride; a.dscale_stride = dscale_stride; a.rgb_scale = rgb_scale;
    a.B = B; a.C = C; a.H = H; a.W = W;
    return OODGAN_OK;
}

}  // namespace

extern "C" int oodgan_act_bwd_sform_nparts(int H, int W) { return (int)(((long)H * W + kP1Chunk - 1) / kP1Chunk); }

extern "C" int oodgan_act_bwd_sform(const float* g_feat, const float* out, const float* noise, int noise_batch,
                                    const float* noise_w, const float* bias, const float* g_rgb, const float* w_rgb,
                                    const float* s_rgb, int s_rgb_stride, float rgb_scale, const float* dscale, int dscale_stride,
                                    const float* mul2, void* ys, float* part_r, float* part_t, float* part_max, int B, int C,
                                    int H, int W, void* stream) {
    ActArgs a;
    int rc = fill_args(a, g_feat, out, noise, noise_batch, noise_w, bias, g_rgb, w_rgb, s_rgb, s_rgb_stride, rgb_scale, dscale,
                       dscale_stride, mul2, part_r, part_t, part_max, B, C, H, W);
    if (rc != OODGAN_OK) return rc;
    OODGAN_REQUIRE(ys != nullptr, "act_bwd_sform: null output");
    const SDims yd = sform_dims(C, H, W);
    a.nparts = oodgan_act_bwd_sform_nparts(H, W);
    OODGAN_REQUIRE((long)B * yd.KC <= 65535, "act_bwd_sform: B*C too large");
    hipLaunchKernelGGL(act_bwd_sform_kernel, dim3(a.nparts, B * yd.KC), dim3(256), 0, as_stream(stream), a,
                       reinterpret_cast<uint4*>(ys), yd);
    return check_launch("act_bwd_sform");
}

extern "C" int oodgan_act_bwd_blurT_nparts(int H, int W) { return ((W + 1 + 31) / 32) * ((H + 1 + 3) / 4); }

extern "C" int oodgan_act_bwd_blurT_sform_phases(const float* g_feat, const float* out, const float* noise, int noise_batch,
                                                 const float* noise_w, const float* bias, const float* g_rgb,
                                                 const float* w_rgb, const float* s_rgb, int s_rgb_stride, float rgb_scale,
                                                 const float* dscale, int dscale_stride, const float* mul2,
                                                 const float* kernel, void* out_phases, float* part_r, float* part_t,
                                                 float* part_max, int B, int C, int H, int W, void* stream) {
    ActArgs a;
    int rc = fill_args(a, g_feat, out, noise, noise_batch, noise_w, bias, g_rgb, w_rgb, s_rgb, s_rgb_stride, rgb_scale, dscale,
                       dscale_stride, mul2, part_r, part_t, part_max, B, C, 2 * H, 2 * W);
    if (rc != OODGAN_OK) return rc;
    OODGAN_REQUIRE(kernel && out_phases, "act_bwd_blurT: null tensor");
    OODGAN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0 && (!g_feat || (reinterpret_cast<uintptr_t>(g_feat) & 15) == 0),
                   "act_bwd_blurT: unaligned input");
    const SPDims d = sp_dims(C, H, W);
    const int tiles_x = (W + 1 + 31) / 32, tiles_y = (H + 1 + 3) / 4;
    a.nparts = tiles_x * tiles_y;
    const long nb = (long)tiles_x * tiles_y * d.KC * B;
    OODGAN_REQUIRE(nb < (1L << 31), "act_bwd_blurT: grid too large");
    if (g_rgb)
        hipLaunchKernelGGL(act_bwd_blurT_sp_kernel<true>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), a, kernel,
                           reinterpret_cast<uint4*>(out_phases), H, W, d, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL(act_bwd_blurT_sp_kernel<false>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), a, kernel,
                           reinterpret_cast<uint4*>(out_phases), H, W, d, tiles_x, tiles_y);
    return check_launch("act_bwd_blurT_sform_phases");
}

extern "C" int oodgan_absmax_scale_check(const float* part, long n, float* state, int* flag, void* stream) {
    OODGAN_REQUIRE(part && state && flag && n > 0, "absmax_scale_check: bad args");
    hipLaunchKernelGGL(absmax_scale_check_kernel, dim3(1), dim3(1024), 0, as_stream(stream), part, n, state, flag);
    return check_launch("absmax_scale_check");
}
