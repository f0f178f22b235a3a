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
#include <type_traits>
#include <cstdint>
#include <cstdlib>

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


// per-block channel constants in LDS: [0] w0 [1] w1 [2] w2 (already x rgb_scale) [3] s_rgb [4] bias [5] d*scale [6] |d|
// fetch_const(): the raw value behind cst[t >> 4][t & 15] for thread t < 112 — ONE unconditional load per thread from a selected,
// clamped address (part_r: a valid address for the constants that do not exist); a switch over `which` with a load in every arm is
// one round trip per arm and wave.  put_const() fixes it up (selects only) and stores it; a kernel may request other data in
// between — the value is not touched before put_const().
struct ConstSel { int which, c; bool has; };
__device__ __forceinline__ ConstSel const_sel(const ActArgs& a, int kc) {
    const int t = threadIdx.x;
    ConstSel s;
    s.which = t >> 4;
    s.c = kc * 16 + (t & 15);
    s.has = s.which < 4 ? a.g_rgb != nullptr : s.which == 4 ? a.bias != nullptr : a.dscale != nullptr;
    return s;
}

__device__ __forceinline__ float fetch_const(const ActArgs& a, int b, int kc) {
    float raw = 0.f;
    if (threadIdx.x < 112) {
        const ConstSel s = const_sel(a, kc);
        const int cc = min(s.c, a.C - 1);
        const float* src = a.part_r;
        if (s.has) {
            if (s.which < 3) src = a.w_rgb + s.which * a.C + cc;
            else if (s.which == 3) src = a.s_rgb + (long)b * a.s_rgb_stride + cc;
            else if (s.which == 4) src = a.bias + cc;
            else src = a.dscale + (long)b * a.dscale_stride + cc;
        }
        raw = *src;
    }
    return raw;
}

__device__ __forceinline__ void put_const(const ActArgs& a, int kc, float (*cst)[16], float raw) {
    const int t = threadIdx.x;
    if (t < 112) {
        const ConstSel s = const_sel(a, kc);
        const float m2 = a.mul2 ? a.mul2[1] : 1.f;
        float v;
        if (s.which < 3) v = s.has ? raw * a.rgb_scale : 0.f;
        else if (s.which <= 4) v = s.has ? raw : 0.f;
        else if (s.which == 5) v = (s.has ? raw : 1.f) * m2;
        else v = s.has ? fabsf(raw) : 1.f;
        cst[t >> 4][t & 15] = s.c < a.C ? v : 0.f;
    }
}

__device__ __forceinline__ void load_consts(const ActArgs& a, int b, int kc, float (*cst)[16]) { put_const(a, kc, cst, fetch_const(a, b, kc)); }

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
    // Phase C: 4 sixteen-byte slots per pixel (slots 0,1 = hi halves of channels 0-7 / 8-15, slots 2,3 = lo halves), one slot
    // task per thread and pass, ordered so that the 64 lanes of a store write 64 CONSECUTIVE slots (1 KB contiguous).  One
    // thread per record would issue its four 16-byte pieces at a 64-byte stride: four times the memory transactions.
#pragma unroll
    for (int rep = 0; rep < kP1Chunk * 4 / 256; ++rep) {
        const int u = tid + rep * 256;
        const int lp = u >> 2, sl = u & 3;
        const long p = p0 + lp;
        if (p >= HW) continue;
        half8 o8;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            const float val = lst[(8 * (sl & 1) + cc) * kP1Pitch + lp];
            const _Float16 hh = (_Float16)val;
            o8[cc] = (sl & 2) ? (_Float16)(val - (float)hh) : hh;
        }
        const int y = (int)(p / a.W), x = (int)(p % a.W);
        reinterpret_cast<half8*>(ys + sform_unit(yd, b, kc, y, x, 0))[sl] = o8;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same producer for an F-form `out` ([B][KC][H*W][16] fp32: oodgan_conv_args.y_fform, written by the strip kernel) and no
// g_feat: the last styled conv.  thread = (pixel, channel quarter): one float4 of `out`, four results, and after one DPP
// exchange with the neighbouring quarter ONE 16-byte slot of the pixel's S-form record — the four lanes of a pixel write its
// 64 bytes in one store instruction, a wave 1 KB contiguous.  No LDS transpose: the NCHW form reads 16 planes and turns them
// through LDS (540 us at 1024² B=8, 4.0 TB/s, 71 % of the wave time waiting).
__global__ __launch_bounds__(256) void act_bwd_sform_f_kernel(const ActArgs a, uint4* __restrict__ ys, const SDims yd) {
    __shared__ float cst[7][16];
    __shared__ float redr[4][16], redt[4][16], redm[4];
    const int KC = yd.KC;
    const int bk = blockIdx.y, b = bk / KC, kc = bk % KC;
    const int tid = threadIdx.x, q = tid & 3, pl = tid >> 2;
    const long HW = (long)a.H * a.W;
    load_consts(a, b, kc, cst);
    __syncthreads();
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const float* np = a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * HW : nullptr;
    const float* gr = a.g_rgb ? a.g_rgb + (long)b * 3 * HW : nullptr;
    const float* of = a.out + ((long)b * KC + kc) * HW * 16 + 4 * q;
    float w0[4], w1[4], w2[4], sr[4], bv[4], ds[4], da[4], acc_r[4], acc_t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w0[j] = cst[0][4 * q + j]; w1[j] = cst[1][4 * q + j]; w2[j] = cst[2][4 * q + j]; sr[j] = cst[3][4 * q + j];
        bv[j] = cst[4][4 * q + j]; ds[j] = cst[5][4 * q + j]; da[j] = cst[6][4 * q + j];
        acc_r[j] = acc_t[j] = 0.f;
    }
    const long p0 = (long)blockIdx.x * kP1Chunk;
    float amax = 0.f;
#pragma unroll 2
    for (int k = 0; k < kP1Chunk / 64; ++k) {
        const long p = p0 + pl + 64 * k;
        const bool ok = p < HW;
        const long pc = ok ? p : HW - 1;
        const float4 o4 = *reinterpret_cast<const float4*>(of + pc * 16);
        const float r0 = gr ? gr[pc] : 0.f, r1 = gr ? gr[HW + pc] : 0.f, r2 = gr ? gr[2 * HW + pc] : 0.f;
        const float nz = np ? nw * np[pc] : 0.f;
        const float ov[4] = {o4.x, o4.y, o4.z, o4.w};
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float o = ov[j];
            const float t = w0[j] * r0 + w1[j] * r1 + w2[j] * r2;
            const float g = sr[j] * t;
            const float gp = ok ? g * (o > 0.f ? kSqrt2 : 0.2f * kSqrt2) : 0.f;
            const float ycv = (o > 0.f ? o * kInvPos : o * kInvNeg) - nz - bv[j];
            acc_r[j] += gp * ycv;
            acc_t[j] += ok ? o * t : 0.f;
            amax = fmaxf(amax, fabsf(gp) * da[j]);
            v[j] = gp * ds[j];
        }
        unsigned h01, l01, h23, l23;
        split_pair(v[0], v[1], h01, l01);
        split_pair(v[2], v[3], h23, l23);
        // quarters (0,1) and (2,3) exchange: the even one collects the hi halves of the 8 channels, the odd one the lo halves
        const bool even = (q & 1) == 0;
        const unsigned s0 = even ? l01 : h01, s1 = even ? l23 : h23;
        const unsigned g0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
        const unsigned g1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, false);
        const uint4 slot = even ? make_uint4(h01, h23, g0, g1) : make_uint4(g0, g1, l01, l23);
        if (ok) {
            const int y = (int)(p / a.W), x = (int)(p % a.W);
            ys[sform_unit(yd, b, kc, y, x, 0) + (even ? (q >> 1) : 2 + (q >> 1))] = slot;
        }
    }
    // per-channel sums: lanes with the same quarter (lane bits 2-5), then the four waves
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
            acc_r[j] += __shfl_xor(acc_r[j], o, 64);
            acc_t[j] += __shfl_xor(acc_t[j], o, 64);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const int lane = tid & 63, wv = tid >> 6;
    if (lane < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { redr[wv][4 * lane + j] = acc_r[j]; redt[wv][4 * lane + j] = acc_t[j]; }
    }
    if (lane == 0) redm[wv] = amax;
    __syncthreads();
    if (tid < 16 && kc * 16 + tid < a.C) {
        const long o = ((long)b * a.C + kc * 16 + tid) * a.nparts + blockIdx.x;
        a.part_r[o] = (redr[0][tid] + redr[1][tid]) + (redr[2][tid] + redr[3][tid]);
        if (a.part_t) a.part_t[o] = (redt[0][tid] + redt[1][tid]) + (redt[2][tid] + redt[3][tid]);
    }
    if (tid == 0) a.part_max[(long)bk * a.nparts + blockIdx.x] = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
}

__global__ __launch_bounds__(256) void from_fform_kernel(const float* __restrict__ f, float* __restrict__ y, int C, long HW, long total) {
    // element e of y (B,C,H,W); F-form [B][KC][HW][16]
    const int KC = (C + 15) / 16;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long p = e % HW;
        const int c = (int)((e / HW) % C);
        const long b = e / (HW * C);
        y[e] = f[((b * KC + (c >> 4)) * HW + p) * 16 + (c & 15)];
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
        const long cbase_c = ((long)b * a.C + min(c, a.C - 1)) * HW;
        const bool has_g = a.g_feat != nullptr, has_n = np != nullptr, has_r = gr != nullptr;
        const float* gsrc = has_g ? a.g_feat : a.out;
        const float* nsrc = has_n ? np : a.out + cbase_c;
        const float* rsrc = has_r ? gr : a.out;                 // (the RGB instance is only launched with g_rgb)
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
                okv[kk] = k0 + kk < NE && act && c < a.C && gy >= 0 && gy < Hg && gx >= 0 && gx + 3 < Wg;
                // unconditional loads from a clamped position (absent tensors: a valid address, masked by has_*): the values are
                // only touched inside `if (okv)` below.  A load inside `if (okv)` is waited for with vmcnt(0) at the merge — the
                // batch was 13 round trips in a row
                const long p = (long)min(max(gy, 0), Hg - 1) * Wg + min(max(gx, 0), Wg - 4);
                o4[kk] = *reinterpret_cast<const float4*>(a.out + cbase_c + p);
                g4[kk] = *reinterpret_cast<const float4*>(gsrc + cbase_c + p);
                nz4[kk] = *reinterpret_cast<const float4*>(nsrc + p);
                if (RGB) {
                    r04[kk] = *reinterpret_cast<const float4*>(rsrc + p);
                    r14[kk] = *reinterpret_cast<const float4*>(rsrc + HW + p);
                    r24[kk] = *reinterpret_cast<const float4*>(rsrc + 2 * HW + p);
                }
            }
#pragma unroll
            for (int kk = 0; kk < NB; ++kk) {
                int r, c4;
                if (k0 + kk >= NE || !tile_elem(k0 + kk, tid & 15, r, c4)) continue;
                const int gy = gy0 + r, gx = gx0 + 4 * c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (okv[kk]) {
                    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 gq = has_g ? g4[kk] : z4, nq = has_n ? nz4[kk] : z4;
                    const float ov[4] = {o4[kk].x, o4[kk].y, o4[kk].z, o4[kk].w}, gv[4] = {gq.x, gq.y, gq.z, gq.w};
                    const float nzv[4] = {nq.x, nq.y, nq.z, nq.w};
                    constexpr int ri = RGB ? 1 : 0;
                    const float4 q0 = (RGB && has_r) ? r04[kk * ri] : z4, q1 = (RGB && has_r) ? r14[kk * ri] : z4, q2 = (RGB && has_r) ? r24[kk * ri] : z4;
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

// ---------------------------------------------------------------------------------------------------------------
// STRIP WALK of the same producer (no ToRGB branch: every up-sampling layer of the generator).
// The tile kernel above fetches an 11 x 72 input tile for every 8 x 64 interior (1.55x, and PMC shows 1.41x of the
// algorithmic bytes reaching HBM) and only loads during the first of its three phases.  Here a workgroup owns
// (b, 16-channel block, 64 g columns = 32 positions) and walks DOWN a segment of position rows:
//   * one iteration = position row i = g2 rows 2i, 2i+1; it consumes the two NEW g rows 2i+1, 2i+2 (72 columns incl. the
//     horizontal halo: 1.125x), whose loads were issued one iteration earlier (register prefetch);
//   * activation gradient on arrival -> LDS row buffer -> the four horizontal 4-tap passes of the row (any 4x4 kernel, no
//     rank-1 assumption) are added to three PENDING output rows kept in registers (no vertical halo re-read, no
//     sliding-window moves: the state is three float4);
//   * scaled results -> LDS [row parity][channel][64] -> one thread gathers 8 channels of a position and writes the hi
//     and the lo 16-byte slot of its 64-byte record.
// Two barriers per iteration; segments of the image height give >= 4 workgroups per CU.  The two warm-up iterations of
// a segment re-read three rows of its upper neighbour.  Partial sums go to the slots of the tile kernel's layout
// (tiles of 4 x 32 positions), one slot per (strip, segment), the segment's other slots are zeroed.

struct StripGeo {
    int nstrips, nseg, seg_rows;      // seg_rows: position rows per segment (>= 16)
    int tiles_x, tiles_y;             // partial-sum slot table of the tile kernel (nstrips <= tiles_x, nseg <= tiles_y)
    int extra;                        // 1: W is a multiple of the strip's positions — the last strip also emits position j = W
    int hi_only;                      // 1 (round 6, oodgan_act_bwd_blurT_sform_phases_hi): 32-byte records, the hi halves only — record r of a phase plane
                                      // at byte r*32 of that plane's FIRST half; the consumer is the two-instruction stride-2 conv (x_hi_only = 2)
};

// QN = float4 column groups per channel: the strip is SW = 4*QN g columns = 2*QN positions wide, the workgroup 16*QN threads.
// Every strip boundary costs two extra 64-byte sectors per row, channel and tensor (the halo loads miss L2: +37 % fetched
// bytes and +25 % time at QN = 16); wider strips did not pay for it (below).
// Measured alternatives (1024² layer, us): 64-column strips x 16 channels (this kernel) 935; 128 columns x 16 channels with
// 512 threads 981-1012; 128 columns x 8 channels (256 threads, half records per workgroup) 1092; no halo loads at all 742.
// Register budget: the hot instance (PRE, rank-one kernel) fits 128 registers = four workgroups per CU; the others (two input tensors and / or a
// general 4x4 kernel: the first W+ step, the exact-scale fallback) spilled 3-36 registers to scratch at 128 / 168 and get 256.
template <int QN, bool XTRA, bool PRE>
__global__ __launch_bounds__(16 * QN) __attribute__((amdgpu_waves_per_eu((PRE && !XTRA) ? 4 : 2, 4))) void act_bwd_blurT_strip_kernel(const ActArgs a, const float* __restrict__ kern, uint4* __restrict__ outp,
                                                                   int H, int W, SPDims sp, StripGeo geo) {
    constexpr int SW = 4 * QN, NT = 16 * QN, NWV = NT / 64;
    constexpr int BS_RP = SW + 4;       // LDS pitch of a g_pre row: 2 + SW + 1 columns, 16-byte aligned rows
    constexpr int BS_GP = SW + 4;       // gather pitch
    static_assert(QN == 16, "a channel row is one DPP row of 16 lanes");
    __shared__ __attribute__((aligned(16))) float gat[2][2][16][BS_GP];      // [iteration parity][g2 row of the pair][channel][column]
    __shared__ float cst[7][16];
    __shared__ float redm[NWV];
    const int tid = threadIdx.x;
    int w;
    {
        const int total = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int strip = w % geo.nstrips; w /= geo.nstrips;
    const int seg = w % geo.nseg; w /= geo.nseg;
    const int kc = w % sp.KC, b = w / sp.KC;
    const int Hg = 2 * H, Wg = 2 * W;
    const long HW = (long)Hg * Wg;
    const int i0 = seg * geo.seg_rows;
    const int i1 = min(i0 + geo.seg_rows, H + 1);
    const int X0 = SW * strip, j0 = (SW / 2) * strip;
    // position j = W (g2 column 2W, the one beyond the last pair of g columns) would need a strip of its own: the last
    // strip emits it as a 65th / 129th output column from the two g columns it already holds
    const bool xtra = XTRA && strip == geo.nstrips - 1;
    const float cval = fetch_const(a, b, kc);               // stored and published behind the first row request (below)
    // the 16 taps (uniform: scalar registers); g2[Y][X] = sum_{a,b} kp[a][b] * g[Y+a-2][X+b-2], kp = the flipped kernel
    float kp[4][4];
#pragma unroll
    for (int aa = 0; aa < 4; ++aa)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) kp[aa][bb] = kern[(3 - aa) * 4 + (3 - bb)];
    // A rank-one kernel (make_kernel of a 1-D filter: kp[a][b] = kv[a] * kh[b]) needs 32 instead of 76 multiply-adds per row and
    // thread — the waves of this kernel are busy 30 % of the time at four per SIMD: it is bound by instruction issue as much as by
    // memory.  Checked here on the 16 values (uniform), so the result does not depend on a promise by the caller.
    float kv[4], kh[4];
    bool rank1 = kp[0][0] != 0.f;
    {
        float kmax = 0.f;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) kmax = fmaxf(kmax, fabsf(kp[aa][bb]));
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            kv[aa] = rank1 ? kp[aa][0] / kp[0][0] : 0.f;
            kh[aa] = kp[0][aa];
        }
#pragma unroll
        for (int aa = 0; aa < 4; ++aa)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) rank1 = rank1 && fabsf(kp[aa][bb] - kv[aa] * kh[bb]) <= 1e-7f * kmax;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa)                       // uniform values: keep them in scalar registers
            kv[aa] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kv[aa])));
    }
    const int ch = tid / QN, q = tid % QN, c = kc * 16 + ch;
    float bv = 0.f, sc_ = 0.f;                               // cst[4][ch], cst[5][ch]: read behind the barrier below
    const int csw = (ch >> 3) & 1;                           // column swizzle of the gather image (see step())
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const float* np = a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * HW : nullptr;
    const long cbase = ((long)b * a.C + c) * HW;
    const bool cok = c < a.C;
    // Columns of a row in LDS: index l = X - X0 + 2, l = 0,1 the left halo, 2..65 the strip's own 64 columns, 66 the right
    // halo.  Thread q owns the aligned float4 X0+4q .. +3 (all of it "owned": counted in the sums); the three halo columns are
    // extras of two threads: q = 0 loads the float2 (X0-2, X0-1), q = QN-1 the single column X0+SW.
    const int gxm = X0 + 4 * q;
    const bool vm = cok && gxm + 3 < Wg;
    const bool vl = cok && q == 0 && X0 >= 2;                    // left pair exists (not the image border)
    const bool vr = cok && q == QN - 1 && X0 + SW < Wg;          // right column exists

    // PRE: g_feat already is g_pre (the conv above applied act' in its epilogue, oodgan_conv_args.dot_actgrad): `out` is not
    // read, and the r sum keeps only its noise / bias term (the caller adds sum dx*out = out_scale * dot of that conv)
    //
    // A load is ONLY a load here.  Every lane reads every piece of a row pair from a clamped address — no branch around a load,
    // no zero written over a loaded value — and what lies outside the image, the channel count or the lane's role is masked in
    // step(), one iteration later, with masks that do not depend on the row (rok: uniform, recomputed from i).  The compiler can
    // only COUNT loads it knows are issued: with the loads of iteration i+1 inside `if (rok && vm)` the wait for iteration i's
    // data came out as vmcnt(0) — the prefetch was waited for right where it was requested (52 % of the wave time waiting at 4.2 TB/s).
    // The halo extras are one float2 per lane: column pair X0-2 for every lane but the last of a channel row, X0+SW for that one.
    struct Rows { float4 o[PRE ? 1 : 2], g[2], n[2]; float2 oe[PRE ? 1 : 2], ge[2]; };    // [row a/b]
    const int gxc = min(gxm, Wg - 4);
    const int ecol = q == QN - 1 ? min(X0 + SW, Wg - 2) : max(X0 - 2, 0);
    const long cbase_c = ((long)b * a.C + min(c, a.C - 1)) * HW;
    const float* gsrc = a.g_feat ? a.g_feat : a.out;        // a valid address when there is no feature gradient (masked below)
    const bool has_g = a.g_feat != nullptr;
    const float* nsrc = np ? np : gsrc + cbase_c;
    const bool has_n = np != nullptr && gxm + 3 < Wg;
    auto load_rows = [&](int i, Rows& R) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int rc = min(max(2 * i + 1 + rr, 0), Hg - 1);
            const long p = cbase_c + (long)rc * Wg;
            if (!PRE) {
                R.o[PRE ? 0 : rr] = *reinterpret_cast<const float4*>(a.out + p + gxc);
                R.oe[PRE ? 0 : rr] = *reinterpret_cast<const float2*>(a.out + p + ecol);
            }
            R.g[rr] = *reinterpret_cast<const float4*>(gsrc + p + gxc);
            R.ge[rr] = *reinterpret_cast<const float2*>(gsrc + p + ecol);
            R.n[rr] = *reinterpret_cast<const float4*>(nsrc + (long)rc * Wg + gxc);
        }
    };

    float acc_r = 0.f, amax = 0.f;
    // pending sums of this thread's 4 columns: before iteration i, pend[0] = g2 row 2i (rows 2i-2..2i in), pend[1] = row 2i+1
    // (rows 2i-1, 2i in), pend[2] = row 2i+2 (row 2i in).  A new g row r adds its four horizontal passes to rows r-1 .. r+2.
    float pend[3][4], pendx[3] = {0.f, 0.f, 0.f};          // pendx: the extra column of the last strip (thread q = QN-1)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) pend[k][e] = 0.f;

    auto step = [&](int i, const Rows& R, auto r1_c) __attribute__((always_inline)) {
        constexpr bool R1 = decltype(r1_c)::value;
        // ---- A/B: activation gradient of the two new rows
        float done[2][4];
        float (*gt)[16][BS_GP] = gat[i & 1];                 // one barrier per iteration: the gather image alternates
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * i + 1 + rr;
            const bool rok = r >= 0 && r < Hg;
            const bool own = r >= 2 * i0 + 1 || (seg == 0 && r == 0);      // warm-up rows belong to the segment above
            const bool mg = rok && vm && has_g, mn = rok && has_n;
            const bool me0 = rok && has_g && (vl || vr), me1 = rok && has_g && vl;      // the right column is the pair's .x
            const float gv[4] = {mg ? R.g[rr].x : 0.f, mg ? R.g[rr].y : 0.f, mg ? R.g[rr].z : 0.f, mg ? R.g[rr].w : 0.f};
            float gp[4], e0, e1;
            const float nv[4] = {mn ? R.n[rr].x : 0.f, mn ? R.n[rr].y : 0.f, mn ? R.n[rr].z : 0.f, mn ? R.n[rr].w : 0.f};
            const float ge0 = me0 ? R.ge[rr].x : 0.f, ge1 = me1 ? R.ge[rr].y : 0.f;
            if (PRE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gp[e] = gv[e];
                    if (own) {
                        acc_r -= gp[e] * (nw * nv[e] + bv);
                        amax = fmaxf(amax, fabsf(gp[e]));
                    }
                }
                e0 = ge0;
                e1 = ge1;
            } else {
                const float4 o4 = R.o[PRE ? 0 : rr];
                const float2 oe = R.oe[PRE ? 0 : rr];
                const bool mo = rok && vm;
                const float ov[4] = {mo ? o4.x : 0.f, mo ? o4.y : 0.f, mo ? o4.z : 0.f, mo ? o4.w : 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = ov[e];
                    gp[e] = gv[e] * (o > 0.f ? kSqrt2 : 0.2f * kSqrt2);
                    if (own) {
                        const float ycv = (o > 0.f ? o * kInvPos : o * kInvNeg) - nw * nv[e] - bv;
                        acc_r += gp[e] * ycv;
                        amax = fmaxf(amax, fabsf(gp[e]));
                    }
                }
                // halo extras (no sums): q = 0 -> l = 0,1; q = 15 -> l = 66
                e0 = ge0 * (oe.x > 0.f ? kSqrt2 : 0.2f * kSqrt2);
                e1 = ge1 * (oe.y > 0.f ? kSqrt2 : 0.2f * kSqrt2);
            }
            // ---- C: the four horizontal 4-tap passes of each new row for columns X0 + 4q .. +3 feed the pending rows; g row 2i+1
            // completes g2 row 2i, g row 2i+2 completes g2 row 2i+1.  Output column e needs columns X0 + 4q + e - 2 .. + 1: the left
            // neighbour's last two values and the right neighbour's first one come by DPP inside the channel's 16 lanes (row_shr:1 /
            // row_shl:1), the strip's halo from the extras of its first and last lane.
            float lz = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gp[2]), 0x111, 0xF, 0xF, false));
            float lw = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gp[3]), 0x111, 0xF, 0xF, false));
            float rx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gp[0]), 0x101, 0xF, 0xF, false));
            if (q == 0) { lz = e0; lw = e1; }
            if (q == QN - 1) rx = e0;
            const float x[7] = {lz, lw, gp[0], gp[1], gp[2], gp[3], rx};
            // row r contributes kp[a] to g2 row r + 2 - a: a = 3 completes pend[0], a = 0 opens a new row
            float dxx = 0.f;
            if constexpr (R1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // explicit fused operations: every instance of the kernel rounds the same way (the tests compare them bit for bit)
                    const float hr = __builtin_fmaf(kh[3], x[e + 3], __builtin_fmaf(kh[2], x[e + 2], __builtin_fmaf(kh[1], x[e + 1], kh[0] * x[e])));
                    done[rr][e] = __builtin_fmaf(kv[3], hr, pend[0][e]);
                    pend[0][e] = __builtin_fmaf(kv[2], hr, pend[1][e]);
                    pend[1][e] = __builtin_fmaf(kv[1], hr, pend[2][e]);
                    pend[2][e] = kv[0] * hr;
                }
                if (XTRA && xtra && q == QN - 1) {
                    const float hr = __builtin_fmaf(kh[1], x[5], kh[0] * x[4]);
                    dxx = __builtin_fmaf(kv[3], hr, pendx[0]);
                    pendx[0] = __builtin_fmaf(kv[2], hr, pendx[1]);
                    pendx[1] = __builtin_fmaf(kv[1], hr, pendx[2]);
                    pendx[2] = kv[0] * hr;
                }
            } else {
                float fresh[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    done[rr][e] = pend[0][e] + (kp[3][0] * x[e] + kp[3][1] * x[e + 1] + kp[3][2] * x[e + 2] + kp[3][3] * x[e + 3]);
                    pend[0][e] = pend[1][e] + (kp[2][0] * x[e] + kp[2][1] * x[e + 1] + kp[2][2] * x[e + 2] + kp[2][3] * x[e + 3]);
                    pend[1][e] = pend[2][e] + (kp[1][0] * x[e] + kp[1][1] * x[e + 1] + kp[1][2] * x[e + 2] + kp[1][3] * x[e + 3]);
                    fresh[e] = kp[0][0] * x[e] + kp[0][1] * x[e + 1] + kp[0][2] * x[e + 2] + kp[0][3] * x[e + 3];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) pend[2][e] = fresh[e];
                if (XTRA && xtra && q == QN - 1) {   // output column X0+SW = 2W: its taps b = 0,1 see the last two g columns, b = 2,3 the border
                    dxx = pendx[0] + (kp[3][0] * x[4] + kp[3][1] * x[5]);
                    pendx[0] = pendx[1] + (kp[2][0] * x[4] + kp[2][1] * x[5]);
                    pendx[1] = pendx[2] + (kp[1][0] * x[4] + kp[1][1] * x[5]);
                    pendx[2] = kp[0][0] * x[4] + kp[0][1] * x[5];
                }
            }
            if (XTRA && xtra && q == QN - 1 && i >= i0) gt[rr][ch][SW ^ csw] = dxx * sc_;
        }
        if (i >= i0) {
            // channels 8-15 keep their columns pairwise swapped (column ^ 1): the gather below reads the two channel halves of a
            // position with lanes 32 banks apart, which the swap turns into the other bank parity
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float d0 = done[rr][0] * sc_, d1 = done[rr][1] * sc_, d2 = done[rr][2] * sc_, d3 = done[rr][3] * sc_;
                *reinterpret_cast<float4*>(&gt[rr][ch][4 * q]) = csw ? make_float4(d1, d0, d3, d2) : make_float4(d0, d1, d2, d3);
            }
        }
        __syncthreads();
        // ---- E: 4 phase rows x SW/2 positions x 2 channel halves = 4*SW tasks, one per thread: a wave is a phase row, a lane pair a
        // record.  The thread converts its 8 channels once and writes the hi slot and the lo slot of its half (two stores of a wave
        // cover bytes 0-31 and 32-63 of 32 consecutive records); with a task per 16-byte slot every value was converted twice.
        if (i >= i0) {
            static_assert(4 * SW == NT, "one gather task per thread");
            const int ph = tid / SW, v = tid % SW;
            const int py = ph >> 1, px = ph & 1, jl = v >> 1, hf = v & 1;
            const int j = j0 + jl, Xl = 2 * jl + px;
            if (j <= W) {
                const bool zero = (2 * i + py > 2 * H) || (X0 + Xl > 2 * W);
                half8 h8, l8;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    float val = gt[py][8 * hf + cc][Xl ^ hf];
                    if (zero) val = 0.f;
                    const _Float16 hh = (_Float16)val;
                    h8[cc] = hh;
                    l8[cc] = (_Float16)(val - (float)hh);
                }
                if (geo.hi_only) {      // a lane pair = the 32 bytes of a record: a wave writes 32 consecutive records = 1 KB contiguous
                    half8* recp = reinterpret_cast<half8*>(outp + ((((long)b * sp.KC + kc) * 4 + ph) * sp.plane + ((long)i * sp.Wq + j) * 2));
                    recp[hf] = h8;
                } else {
                    half8* recp = reinterpret_cast<half8*>(outp + ((((long)b * sp.KC + kc) * 4 + ph) * sp.plane + ((long)i * sp.Wq + j) * 4));
                    recp[hf] = h8;
                    recp[2 + hf] = l8;
                }
            }
            if (XTRA && xtra && tid < 16) {          // the four records of position j = W (px = 1 lies beyond the image: zeros)
                const int ph = tid >> 2, sl = tid & 3, py = ph >> 1, px = ph & 1;
                const bool zero = px == 1 || (2 * i + py > 2 * H);
                half8 o8;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    float val = zero ? 0.f : gt[py][8 * (sl & 1) + cc][SW ^ (sl & 1)];
                    const _Float16 hh = (_Float16)val;
                    o8[cc] = (sl & 2) ? (_Float16)(val - (float)hh) : hh;
                }
                if (geo.hi_only) {
                    if (sl < 2) reinterpret_cast<half8*>(outp + ((((long)b * sp.KC + kc) * 4 + ph) * sp.plane + ((long)i * sp.Wq + W) * 2))[sl] = o8;
                } else {
                    half8* recp = reinterpret_cast<half8*>(outp + ((((long)b * sp.KC + kc) * 4 + ph) * sp.plane + ((long)i * sp.Wq + W) * 4));
                    recp[sl] = o8;
                }
            }
        }
    };

    // two warm-up iterations fill the history (rows 2*i0-3 .. 2*i0), then one iteration per position row; the loads of
    // iteration i+1 are in flight while iteration i computes
    Rows ra, rb;
    load_rows(i0 - 2, ra);
    put_const(a, kc, cst, cval);
    __syncthreads();
    bv = cst[4][ch];
    sc_ = cst[5][ch];
    int i = i0 - 2;
    auto walk = [&](auto r1_c) __attribute__((always_inline)) {
        for (; i + 1 < i1; i += 2) {
            load_rows(i + 1, rb);
            step(i, ra, r1_c);
            load_rows(i + 2, ra);
            step(i + 1, rb, r1_c);
        }
        if (i < i1) step(i, ra, r1_c);
    };
    if (rank1) walk(std::true_type{});
    else walk(std::false_type{});

    // ---- per-channel sums and the block maximum -> the tile kernel's slot layout
#pragma unroll
    for (int o = QN / 2; o > 0; o >>= 1) acc_r += __shfl_xor(acc_r, o, 64);
    // One slot of the tile kernel's (tiles_y x tiles_x) slot table per workgroup: row = segment, column = strip.  The workgroup of the last used row /
    // column also zeroes the unused rows / columns, so that a plain sum / max over the table stays correct.
    auto put = [&](float* tab, int row, int rows_used, float v) {
        const int r1 = row == rows_used - 1 ? geo.tiles_y : row + 1;
        const int c1 = strip == geo.nstrips - 1 ? geo.tiles_x : strip + 1;
        for (int r = row; r < r1; ++r)
            for (int cx = strip; cx < c1; ++cx) tab[(long)r * geo.tiles_x + cx] = (r == row && cx == strip) ? v : 0.f;
    };
    if (q == 0 && cok) put(a.part_r + ((long)b * a.C + c) * a.nparts, seg, geo.nseg, acc_r);
    amax *= cst[6][ch];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((tid & 63) == 0) redm[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) {
        float m = redm[0];
        for (int wv = 1; wv < NWV; ++wv) m = fmaxf(m, redm[wv]);
        put(a.part_max + ((long)b * sp.KC + kc) * a.nparts, seg, geo.nseg, m);
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
    OODGAN_REQUIRE(part_r && part_max && mul2 && B > 0 && C > 0 && H > 0 && W > 0, "act_bwd producer: bad args");
    OODGAN_REQUIRE(out || (g_feat && !g_rgb), "act_bwd producer: without out, g_feat must be the pre-activation gradient (no ToRGB branch)");
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

extern "C" int oodgan_act_bwd_sform_f(const float* out_f, const float* noise, int noise_batch, const float* noise_w, const float* bias,
                                      const float* g_rgb, const float* w_rgb, const float* s_rgb, int s_rgb_stride, float rgb_scale,
                                      const float* dscale, int dscale_stride, const float* mul2, void* ys, float* part_r, float* part_t,
                                      float* part_max, int B, int C, int H, int W, void* stream) {
    ActArgs a;
    int rc = fill_args(a, nullptr, out_f, noise, noise_batch, noise_w, bias, g_rgb, w_rgb, s_rgb, s_rgb_stride, rgb_scale, dscale,
                       dscale_stride, mul2, part_r, part_t, part_max, B, C, H, W);
    if (rc != OODGAN_OK) return rc;
    OODGAN_REQUIRE(ys != nullptr && (C % 16) == 0 && (reinterpret_cast<uintptr_t>(out_f) & 15) == 0,
                   "act_bwd_sform_f: needs ys, C %% 16 == 0 and a 16-byte aligned F-form input");
    const SDims yd = sform_dims(C, H, W);
    a.nparts = oodgan_act_bwd_sform_nparts(H, W);
    OODGAN_REQUIRE((long)B * yd.KC <= 65535, "act_bwd_sform_f: B*C too large");
    hipLaunchKernelGGL(act_bwd_sform_f_kernel, dim3(a.nparts, B * yd.KC), dim3(256), 0, as_stream(stream), a,
                       reinterpret_cast<uint4*>(ys), yd);
    return check_launch("act_bwd_sform_f");
}

extern "C" int oodgan_from_fform(const float* f, float* y, int B, int C, int H, int W, void* stream) {
    OODGAN_REQUIRE(f && y && B > 0 && C > 0 && (C % 16) == 0 && H > 0 && W > 0, "from_fform: bad args");
    const long HW = (long)H * W, total = (long)B * C * HW;
    hipLaunchKernelGGL(from_fform_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), f, y, C, HW, total);
    return check_launch("from_fform");
}

static bool blurT_strip_enabled() { return oodgan::tunable(oodgan::OODGAN_TUN_BLURT_STRIP) != 0; }      // tests set 0 to compare with the tile kernel

// 1 when oodgan_act_bwd_blurT_sform_phases takes out == NULL (g_feat already multiplied by act'(out) by the conv above,
// oodgan_conv_args.dot_actgrad) for an up-conv input of H x W: the strip walk only
extern "C" int oodgan_act_bwd_blurT_pre_supported(int H, int W) { return H >= 32 && W >= 32 && blurT_strip_enabled() ? 1 : 0; }

extern "C" int oodgan_act_bwd_blurT_nparts(int H, int W) { return ((W + 1 + 31) / 32) * ((H + 1 + 3) / 4); }

static int act_bwd_blurT_impl(const float* g_feat, const float* out, const float* noise, int noise_batch, const float* noise_w, const float* bias,
                              const float* g_rgb, const float* w_rgb, const float* s_rgb, int s_rgb_stride, float rgb_scale, const float* dscale,
                              int dscale_stride, const float* mul2, const float* kernel, void* out_phases, float* part_r, float* part_t, float* part_max,
                              int B, int C, int H, int W, void* stream, int hi_only);

// 1 when oodgan_act_bwd_blurT_sform_phases_hi exists for an up-conv input of H x W: the strip walk only (no ToRGB branch)
extern "C" int oodgan_act_bwd_blurT_hi_supported(int H, int W) { return H >= 32 && W >= 32 && blurT_strip_enabled() ? 1 : 0; }

extern "C" int oodgan_act_bwd_blurT_sform_phases(const float* g_feat, const float* out, const float* noise, int noise_batch,
                                                 const float* noise_w, const float* bias, const float* g_rgb,
                                                 const float* w_rgb, const float* s_rgb, int s_rgb_stride, float rgb_scale,
                                                 const float* dscale, int dscale_stride, const float* mul2,
                                                 const float* kernel, void* out_phases, float* part_r, float* part_t,
                                                 float* part_max, int B, int C, int H, int W, void* stream) {
    return act_bwd_blurT_impl(g_feat, out, noise, noise_batch, noise_w, bias, g_rgb, w_rgb, s_rgb, s_rgb_stride, rgb_scale, dscale, dscale_stride, mul2, kernel,
                              out_phases, part_r, part_t, part_max, B, C, H, W, stream, 0);
}

extern "C" int oodgan_act_bwd_blurT_sform_phases_hi(const float* g_feat, const float* out, const float* noise, int noise_batch,
                                                    const float* noise_w, const float* bias, const float* dscale, int dscale_stride,
                                                    const float* mul2, const float* kernel, void* out_phases, float* part_r, float* part_max,
                                                    int B, int C, int H, int W, void* stream) {
    OODGAN_REQUIRE(oodgan_act_bwd_blurT_hi_supported(H, W), "act_bwd_blurT hi-only records: the strip walk only (H, W >= 32, tunable blurt_strip)");
    return act_bwd_blurT_impl(g_feat, out, noise, noise_batch, noise_w, bias, nullptr, nullptr, nullptr, 0, 0.f, dscale, dscale_stride, mul2, kernel,
                              out_phases, part_r, nullptr, part_max, B, C, H, W, stream, 1);
}

static int act_bwd_blurT_impl(const float* g_feat, const float* out, const float* noise, int noise_batch, const float* noise_w, const float* bias,
                              const float* g_rgb, const float* w_rgb, const float* s_rgb, int s_rgb_stride, float rgb_scale, const float* dscale,
                              int dscale_stride, const float* mul2, const float* kernel, void* out_phases, float* part_r, float* part_t, float* part_max,
                              int B, int C, int H, int W, void* stream, int hi_only) {
    ActArgs a;
    int rc = fill_args(a, g_feat, out, noise, noise_batch, noise_w, bias, g_rgb, w_rgb, s_rgb, s_rgb_stride, rgb_scale, dscale,
                       dscale_stride, mul2, part_r, part_t, part_max, B, C, 2 * H, 2 * W);
    if (rc != OODGAN_OK) return rc;
    OODGAN_REQUIRE(kernel && out_phases, "act_bwd_blurT: null tensor");
    OODGAN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0 && (!g_feat || (reinterpret_cast<uintptr_t>(g_feat) & 15) == 0),
                   "act_bwd_blurT: unaligned input");
    OODGAN_REQUIRE(out || oodgan_act_bwd_blurT_pre_supported(H, W), "act_bwd_blurT: the pre-activated form (out == NULL) exists only in the "
                   "strip walk (H, W >= 32)");
    const SPDims d = sp_dims(C, H, W);
    const int tiles_x = (W + 1 + 31) / 32, tiles_y = (H + 1 + 3) / 4;
    a.nparts = tiles_x * tiles_y;
    const long nb = (long)tiles_x * tiles_y * d.KC * B;
    OODGAN_REQUIRE(nb < (1L << 31), "act_bwd_blurT: grid too large");
    // strip walk (no ToRGB branch; every up-sampling layer of the generator): enough rows to amortise the two warm-up
    // iterations of a segment
    if (!g_rgb && !part_t && H >= 32 && W >= 32 && blurT_strip_enabled()) {
        constexpr int QN = 16;
        StripGeo geo;
        // the extra-column form saves the (W+1)-th strip and costs registers in every workgroup (20 spilled): it pays on the
        // narrowest layer only (64²: 87 -> 75 us; 128² and 256²: 139 -> 146, 252 -> 275)
        geo.hi_only = hi_only;
        geo.extra = ((W % (2 * QN)) == 0 && W <= 32) ? 1 : 0;
        geo.nstrips = geo.extra ? W / (2 * QN) : (W + 1 + 2 * QN - 1) / (2 * QN);
        geo.tiles_x = tiles_x;
        geo.tiles_y = tiles_y;
        // about four rounds of the 1024 workgroups the chip holds (4 x 256 threads per CU; 1088 workgroups would run as two
        // rounds, the second 6 % full); segments of at least 16 position rows (two warm-up iterations each)
        const long base = (long)B * d.KC * geo.nstrips;
        long nseg = (4L * 1024) / base;
        if (nseg < 1) nseg = 1;
        int seg_rows = (int)((H + 1 + nseg - 1) / nseg);
        if (seg_rows < 16) seg_rows = 16;
        geo.seg_rows = seg_rows;
        geo.nseg = (H + 1 + seg_rows - 1) / seg_rows;
        OODGAN_REQUIRE(geo.nseg <= tiles_y && geo.nstrips <= tiles_x, "act_bwd_blurT strip: slot table too small");
        const long nbs = base * geo.nseg;
#define OODGAN_STRIP_LAUNCH(X, P)                                                                                              \
    hipLaunchKernelGGL((act_bwd_blurT_strip_kernel<QN, X, P>), dim3((unsigned)nbs), dim3(256), 0, as_stream(stream), a, kernel, \
                       reinterpret_cast<uint4*>(out_phases), H, W, d, geo)
        if (geo.extra) {
            if (out) OODGAN_STRIP_LAUNCH(true, false); else OODGAN_STRIP_LAUNCH(true, true);
        } else {
            if (out) OODGAN_STRIP_LAUNCH(false, false); else OODGAN_STRIP_LAUNCH(false, true);
        }
#undef OODGAN_STRIP_LAUNCH
        return check_launch("act_bwd_blurT_sform_phases/strip");
    }
    OODGAN_REQUIRE(!hi_only, "act_bwd_blurT hi-only records: not reachable (strip walk refused this call)");
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
