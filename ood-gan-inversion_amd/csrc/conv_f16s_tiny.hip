// Split-f16 3x3 convs of the 4x4 and 8x8 layers (reference ModulatedConv2d.forward on the constant input and the first up-sampling
// level, src/ops/StyleGAN/model.py:233-274, and their input gradients: stride 1 on an S-form input, stride 2 on a phase-split
// S-form input).  These layers are all weight: 512 x 512 x 9 x {hi,lo} = 9.4 MB against 0.6 GFLOP, and the tile kernels spend a
// whole 8 x 32-position tile per image on 16 or 64 pixels (conv_f16s_s1ring_kernel: 66 us, conv_f16s_s2v2_kernel: 123 us per
// launch, 128 workgroups, at most one per CU, every one of them streaming all weights of its 32 output channels).
// Here the conv is a skinny GEMM: N = B x H x W positions (128 or 512 at B = 8) PACKED into 32-wide N tiles across the images,
// M = output channels in blocks of 32, K = 9 x input channels cut into KS slices:
//   * grid = (M blocks) x (groups of four N tiles) x KS; a workgroup's four waves split its K slice again; a wave keeps
//     4 N tiles x 2 accumulation chains and streams, per K step (16 channels x one tap), the {hi,lo} A fragments of its 32 rows
//     and the {hi,lo} B fragments of its four N tiles straight from L2 into registers (no LDS, no DMA: 590 KB of x, every
//     weight read exactly once per N group), one step ahead of the matrix instructions;
//   * the four waves add their partial tiles through LDS and every workgroup writes its partial (16 KB) to a workspace; a second
//     launch adds the KS partials of a tile in fixed order (deterministic) and runs the epilogue: unscale, out_scale, noise +
//     bias + lrelu*sqrt2 (forward) or the style-gradient dot with the saved forward input (input gradient), NCHW stores.
// The caller provides the workspace (`oodgan_conv_args.workspace`, oodgan_conv3x3_tiny_workspace bytes) — one per HIP stream that
// runs these layers concurrently.
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

struct SPDimsT { int KC, Hq, Wq; long plane; };
__host__ __device__ inline SPDimsT sp_dims_t(int C, int H, int W) {    // must match conv_f16s_v2.hip / bwd_producers.hip
    SPDimsT d;
    d.KC = (C + 15) / 16;
    d.Hq = (H + 7) / 8 * 8 + 2;
    d.Wq = (W + 31) / 32 * 32 + 2;
    d.plane = (long)d.Hq * d.Wq * 4;
    return d;
}

constexpr int TY_NT = 4;                 // N tiles per workgroup
constexpr int TY_STEPS = 5;              // K steps per wave (all of them in registers at once)
constexpr int TY_OP = 132;               // LDS pitch of the epilogue image [32 channels][128 positions]

struct TinyArgs {
    oodgan_conv_args a;
    const float* w_unscale;
    int H, npix, N, ngroups, mblocks, KS, Mp, nsteps;      // H = output height = width; nsteps = 9 * K / 16
    SDims xd;                // mode S1
    SPDimsT sp;              // mode S2
    float* ws;               // [tile][KS][4096] partial tiles
};

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_f16s_tiny_kernel(const TinyArgs p, const uint4* __restrict__ wpk16) {
    __shared__ __attribute__((aligned(16))) float lred[2][TY_NT * 16 * 64];     // 32 KB: partial tiles of two waves at a time
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    int w = blockIdx.x;
    const int ks = w % p.KS; w /= p.KS;
    const int ngrp = w % p.ngroups;
    const int mblk = w / p.ngroups;
    const int tile = mblk * p.ngroups + ngrp;
    const int m0 = mblk * 32;

    // this lane's position in each of the four N tiles: n -> (image, pixel) -> 16-byte unit of the record's slot 0 at tap (0,0)
    long xb[TY_NT];
#pragma unroll
    for (int nt = 0; nt < TY_NT; ++nt) {
        int n = (ngrp * TY_NT + nt) * 32 + l31;
        if (n >= p.N) n = 0;                                 // results of positions past the end are never stored
        const int b = n / p.npix, pix = n % p.npix, y = pix / p.H, x = pix % p.H;
        if (MODE == OODGAN_CONV_S1) xb[nt] = ((long)b * p.xd.KC * p.xd.Hp + y) * p.xd.Wp * 4 + (long)x * 4;      // padded (y + dy, x + dx), dy, dx = 0..2
        else xb[nt] = ((long)b * p.sp.KC * 4) * p.sp.plane + ((long)y * p.sp.Wq + x) * 4;                           // G_ph[i + a][j + b]
    }
    const uint4* xs = reinterpret_cast<const uint4*>(a.x);
    // grouped convolution (oodgan_conv_args.groups): the channel block selects its group's K/16 chunks of the G*K/16 the S-form holds
    const int gk = a.groups > 1 ? (m0 / (a.M / a.groups)) * (a.K / 16) : 0;
    const int slot_hi = half, slot_lo = 2 + half;            // channels 8*half .. +7 of the 16-channel block: hi / lo slot

    // K steps of this wave: the workgroup's slice [ks*per, ...) cut in four
    const int per_wg = (p.nsteps + p.KS - 1) / p.KS;
    const int q0w = ks * per_wg, q1w = min(q0w + per_wg, p.nsteps);
    const int per_wave = (q1w - q0w + 3) / 4;
    const int q0 = min(q0w + wave * per_wave, q1w), q1 = min(q0 + per_wave, q1w);

    auto load_step = [&](int q, half8& ahi, half8& alo, half8 (&bhi)[TY_NT], half8 (&blo)[TY_NT]) {
        const int kc = q / 9, tap = q % 9;
        const half8* wb = reinterpret_cast<const half8*>(wpk16) + ((long)(kc * 9 + tap) * 4) * p.Mp + m0 + l31;
        ahi = wb[(0 * 2 + half) * p.Mp];
        alo = wb[(1 * 2 + half) * p.Mp];
        long off;
        if (MODE == OODGAN_CONV_S1) {
            off = (long)(kc + gk) * p.xd.plane + ((long)(tap / 3) * p.xd.Wp + (tap % 3)) * 4;
        } else {
            const int ky = tap / 3, kx = tap % 3;
            off = ((long)(kc + gk) * 4 + (ky & 1) * 2 + (kx & 1)) * p.sp.plane + ((long)(ky >> 1) * p.sp.Wq + (kx >> 1)) * 4;
        }
#pragma unroll
        for (int nt = 0; nt < TY_NT; ++nt) {
            const half8* xp = reinterpret_cast<const half8*>(xs + xb[nt] + off);
            bhi[nt] = xp[slot_hi];
            blo[nt] = xp[slot_lo];
        }
    };

    f32x16 acc0[TY_NT], acc1[TY_NT];                         // hi*hi ; hi*lo + lo*hi
#pragma unroll
    for (int nt = 0; nt < TY_NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[nt][r] = acc1[nt][r] = 0.f;
    // A wave's whole K share (at most TY_STEPS steps x 10 fragments) is requested before the first matrix instruction: these
    // layers are bound by memory LATENCY (9.4 MB of weights that nobody re-reads, ~0.2 us of matrix work per step), so the loads of
    // all steps must be in flight together — one step of prefetch made every step a full round trip (2 us).
    {
        half8 ahi[TY_STEPS], alo[TY_STEPS], bhi[TY_STEPS][TY_NT], blo[TY_STEPS][TY_NT];
#pragma unroll
        for (int i = 0; i < TY_STEPS; ++i) load_step(min(q0 + i, p.nsteps - 1), ahi[i], alo[i], bhi[i], blo[i]);
#pragma unroll
        for (int i = 0; i < TY_STEPS; ++i) {
            if (q0 + i < q1) {
#pragma unroll
                for (int nt = 0; nt < TY_NT; ++nt) {
                    acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], bhi[i][nt], acc0[nt], 0, 0, 0);
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[i], blo[i][nt], acc1[nt], 0, 0, 0);
                }
#pragma unroll
                for (int nt = 0; nt < TY_NT; ++nt) acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[i], bhi[i][nt], acc1[nt], 0, 0, 0);
            }
        }
    }
    // ---- the four waves' partial tiles -> one per workgroup (element e = (nt*16 + r)*64 + lane): waves 2, 3 hand theirs to waves 0, 1
    if (wave >= 2) {
#pragma unroll
        for (int nt = 0; nt < TY_NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) lred[wave - 2][(nt * 16 + r) * 64 + lane] = acc0[nt][r] + acc1[nt][r];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int nt = 0; nt < TY_NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* c = &lred[wave][(nt * 16 + r) * 64 + lane];
                *c = (acc0[nt][r] + acc1[nt][r]) + *c;
            }
    }
    __syncthreads();
    float* wsp = p.ws + ((long)tile * p.KS + ks) * (TY_NT * 16 * 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i;
        wsp[e] = lred[0][e] + lred[1][e];
    }
}

// Second launch: the KS partial tiles of an (M block, N group) added in the order of k (deterministic) and the epilogue.  A
// workgroup takes a QUARTER of the tile — channels 8q .. 8q+7, all 128 positions: elements (nt*16 + 4q + rr)*64 + lane — so that
// every partial is one coalesced load per thread and all KS x 4 of them are in flight together.  (A single launch with an
// arrival counter and the last workgroup finishing the tile was 2-4x slower: the agent-scope fences it needs between workgroups
// of different XCDs write back and invalidate L2 in every workgroup.)
__global__ __launch_bounds__(256) void conv_f16s_tiny_finish_kernel(const TinyArgs p) {
    __shared__ float lo[8 * TY_OP];
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, rr = tid >> 6;
    const int qd = blockIdx.x & 3, tile = blockIdx.x >> 2;
    const int ngrp = tile % p.ngroups, mblk = tile / p.ngroups;
    const float* wst = p.ws + (long)tile * p.KS * (TY_NT * 16 * 64);
    // the partials of eight K slices in flight together (clamped slice index, masked by a select), added in the order of k as before:
    // one slice per trip was a round trip per slice — 15 in a row in a kernel of 9 us
    float v[TY_NT] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.KS; k0 += 8) {
        float t[8][TY_NT];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int nt = 0; nt < TY_NT; ++nt)
                t[u][nt] = wst[(long)min(k0 + u, p.KS - 1) * (TY_NT * 16 * 64) + (nt * 16 + 4 * qd + rr) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int nt = 0; nt < TY_NT; ++nt) v[nt] += k0 + u < p.KS ? t[u][nt] : 0.f;
    }
    // r = 4 qd + rr -> channel (r & 3) + 8 (r >> 2) + 4 half = rr + 8 qd + 4 half: local channel rr + 4 half of the quarter
#pragma unroll
    for (int nt = 0; nt < TY_NT; ++nt) lo[(rr + 4 * (lane >> 5)) * TY_OP + nt * 32 + (lane & 31)] = v[nt];
    __syncthreads();
    // thread = (local channel cl = tid >> 5, chunk of 4 positions = tid & 31): an image's positions are npix / 4 adjacent chunks
    const int cl = tid >> 5, chunk = tid & 31, m = mblk * 32 + 8 * qd + cl;
    const int n0 = ngrp * (TY_NT * 32) + chunk * 4;
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
    float dsum = 0.f;
    const bool live = n0 < p.N && m < a.M;
    if (p.H < 4) {
        // 1x1 / 2x2 outputs (the last steps of the e4e style heads): a chunk of four positions spans rows or images — every position
        // on its own; no noise, no dot (tiny_eligible)
        if (m < a.M) {
            const int pitch = a.out_pitch ? a.out_pitch : p.H;
            const float bv = a.bias ? a.bias[m] : 0.f, sl = a.act == OODGAN_ACT_PRELU ? a.slope[m] : 0.2f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + j;
                if (n < p.N) {
                    const int b = n / p.npix, pix = n % p.npix;
                    float o = lo[cl * TY_OP + chunk * 4 + j] * us * (a.out_scale ? a.out_scale[(long)b * a.out_scale_stride + m] : 1.f) + bv;
                    if (a.act == OODGAN_ACT_LRELU) o = (o > 0.f ? o : 0.2f * o) * kSqrt2;
                    else if (a.act == OODGAN_ACT_PRELU) o = o > 0.f ? o : sl * o;
                    a.y[((long)b * a.M + m) * ((long)p.H * pitch) + (pix / p.H) * pitch + pix % p.H] = o;
                }
            }
        }
    } else {
        // every thread loads (clamped chunk and channel; the workspace as a valid address for an absent tensor), one round trip for
        // the four inputs; only live threads use and store
        const int n0c = min(n0, p.N - 4), mc = min(m, a.M - 1);
        const int b = n0c / p.npix, pix0 = n0c % p.npix;
        const float scl = *(a.out_scale ? a.out_scale + (long)b * a.out_scale_stride + mc : p.ws);
        const float bvl = *(a.bias ? a.bias + mc : p.ws);
        const float slp = *(a.act == OODGAN_ACT_PRELU ? a.slope + mc : p.ws);
        const float4 nz4 = *reinterpret_cast<const float4*>(a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * p.npix + pix0 : p.ws);
        const float4 d4 = *reinterpret_cast<const float4*>(a.dotx ? a.dotx + ((long)b * a.M + mc) * p.npix + pix0 : p.ws);
        const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
        const float nzv[4] = {nz4.x, nz4.y, nz4.z, nz4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
        if (live) {
            const float sc1 = a.out_scale ? scl : 1.f, bv = a.bias ? bvl : 0.f;
            const int pitch = a.out_pitch ? a.out_pitch : p.H;
            float* yb = a.y ? a.y + ((long)b * a.M + m) * ((long)p.H * pitch) : nullptr;
            const int py = pix0 / p.H, px0 = pix0 % p.H;     // 4 | H: a chunk lies inside one row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = lo[cl * TY_OP + chunk * 4 + j] * us;
                if (a.dotx) dsum += t * dv[j];
                if (yb) {
                    float o = t * sc1 + (a.noise ? nw * nzv[j] : 0.f) + bv;
                    if (a.act == OODGAN_ACT_LRELU) o = (o > 0.f ? o : 0.2f * o) * kSqrt2;
                    else if (a.act == OODGAN_ACT_PRELU) o = o > 0.f ? o : slp * o;
                    yb[py * pitch + px0 + j] = o;
                }
            }
        }
    }
    if (a.dotx) {
        dsum += __shfl_xor(dsum, 1, 64);
        dsum += __shfl_xor(dsum, 2, 64);
        if (p.npix == 64) {
            dsum += __shfl_xor(dsum, 4, 64);
            dsum += __shfl_xor(dsum, 8, 64);
        }
        if (live && (chunk & (p.npix / 4 - 1)) == 0) a.dot_part[((long)(n0 / p.npix) * a.M + m) * a.dot_nparts] = dsum;
    }
}

bool tiny_shape(const oodgan_conv_args& a, int& H) {
    if (a.mode == OODGAN_CONV_S1) H = a.Hin;
    else if (a.mode == OODGAN_CONV_S2) H = (a.Hin - 1) / 2;
    else return false;
    const int Wd = a.mode == OODGAN_CONV_S1 ? a.Win : (a.Win - 1) / 2;
    // 1x1 / 2x2: only the stride-2 chains of the style heads end there (no noise, no dot: tiny_eligible)
    // 16x16 / 32x32 stride-1 maps with at most 1024 positions in all: the encoder trunk at batch 1-4, where the tile kernels have 8-32
    // workgroups walking their K chunks one after the other (the caller opts in by passing a workspace)
    const bool mid = a.mode == OODGAN_CONV_S1 && (H == 16 || H == 32) && a.B * H * H <= tunable(OODGAN_TUN_TINY_MID_MAX);
    return a.x_sform && H == Wd && (H == 4 || H == 8 || mid || (a.mode == OODGAN_CONV_S2 && (H == 1 || H == 2))) && a.K % 16 == 0 && a.K >= 64 &&
           a.M % 32 == 0 && a.B * H * H <= 16384;
}

}  // namespace

namespace oodgan {

// true when the skinny-GEMM kernel takes this call: 4x4 / 8x8 outputs, S-form (mode S1) or phase-split S-form (mode S2) input, a
// workspace from the caller, none of the options the big layers' kernels have (S-form output, fused ToRGB / activation backward)
bool tiny_eligible(const oodgan_conv_args& a) {
    int H;
    return a.workspace != nullptr && tiny_shape(a, H) && a.ys == nullptr && a.rgb_y == nullptr && a.fuse == nullptr && !a.dot_actgrad &&
           !a.y_fform && !a.x_fform && a.in_scale == nullptr && a.in_shift == nullptr &&
           (a.groups <= 1 || (a.mode == OODGAN_CONV_S2 && a.M % a.groups == 0 && (a.M / a.groups) % 32 == 0)) &&
           (a.act == OODGAN_ACT_NONE || a.act == OODGAN_ACT_LRELU || (a.act == OODGAN_ACT_PRELU && a.slope != nullptr)) &&
           (a.y != nullptr || a.dotx != nullptr) && (a.dotx == nullptr || a.dot_nparts == 1) &&
           (H >= 4 || (a.noise == nullptr && a.dotx == nullptr && a.y != nullptr));
}

static void tiny_split(const oodgan_conv_args& a, int H, int& ngroups, int& mblocks, int& KS) {
    const int N = a.B * H * H, ntile = (N + 31) / 32;
    ngroups = (ntile + TY_NT - 1) / TY_NT;
    mblocks = a.M / 32;
    const int nsteps = 9 * (a.K / 16);
    // every wave takes at most TY_STEPS K steps: KS = ceil(nsteps / (4 waves x TY_STEPS))
    KS = (nsteps + 4 * TY_STEPS - 1) / (4 * TY_STEPS);
    if (KS < 1) KS = 1;
}

int launch_tiny(const oodgan_conv_args& a, const void* wpk16, const float* unscale, hipStream_t st) {
    TinyArgs p;
    p.a = a;
    p.w_unscale = unscale;
    int H;
    OODGAN_REQUIRE(tiny_shape(a, H), "conv3x3 tiny: shape");
    p.H = H; p.npix = H * H; p.N = a.B * p.npix;
    tiny_split(a, H, p.ngroups, p.mblocks, p.KS);
    p.Mp = (a.M + 63) / 64 * 64;
    p.nsteps = 9 * (a.K / 16);
    const int G = a.groups > 1 ? a.groups : 1;
    p.xd = sform_dims(a.K * G, a.Hin, a.Win);
    p.sp = sp_dims_t(a.K * G, H, H);
    const long ntiles = (long)p.ngroups * p.mblocks;
    OODGAN_REQUIRE((p.nsteps + p.KS - 1) / p.KS <= 4 * TY_STEPS, "conv3x3 tiny: K split");
    OODGAN_REQUIRE(a.workspace_bytes >= ntiles * p.KS * (TY_NT * 16 * 64) * 4, "conv3x3 tiny: workspace too small (oodgan_conv3x3_tiny_workspace)");
    OODGAN_REQUIRE(!a.dotx || a.dot_part, "conv3x3: dotx without dot_part");
    p.ws = reinterpret_cast<float*>(a.workspace);
    const dim3 grid((unsigned)(ntiles * p.KS)), block(256);
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
    if (a.mode == OODGAN_CONV_S1) hipLaunchKernelGGL((conv_f16s_tiny_kernel<OODGAN_CONV_S1>), grid, block, 0, st, p, w16);
    else hipLaunchKernelGGL((conv_f16s_tiny_kernel<OODGAN_CONV_S2>), grid, block, 0, st, p, w16);
    hipLaunchKernelGGL(conv_f16s_tiny_finish_kernel, dim3((unsigned)(ntiles * 4)), block, 0, st, p);
    return check_launch("conv3x3_f16s_tiny");
}

}  // namespace oodgan

// bytes of workspace (the K-split partial tiles) the skinny-GEMM kernel needs for this call, 0 when it does not apply
extern "C" long oodgan_conv3x3_tiny_workspace(int mode, int B, int K, int M, int Hin, int Win) {
    oodgan_conv_args a = {};
    a.mode = mode; a.B = B; a.K = K; a.M = M; a.Hin = Hin; a.Win = Win; a.x_sform = 1;
    int H;
    if (!tiny_shape(a, H)) return 0;
    int ng, mb, KS;
    oodgan::tiny_split(a, H, ng, mb, KS);
    return (long)ng * mb * KS * (TY_NT * 16 * 64) * 4;
}
