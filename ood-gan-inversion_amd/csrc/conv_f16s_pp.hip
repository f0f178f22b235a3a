// Split-f16 3x3 stride-1 conv, "ping-pong" structure: the plain ModulatedConv2d forward, its input gradient and the
// dense AlignNet convs — the kernel that carries most of the path's FLOPs.
//
// Same arithmetic and LDS images as conv_f16s.hip (v = hi + lo in f16, three v_mfma_f32_32x32x16_f16 per product,
// fp32 accumulate).  What changes is the schedule: a 512-thread workgroup holds TWO independent 4-wave pipelines
// ("groups"), each owning one output tile (64 channels x 8x32 pixels) and its own LDS buffers, offset in time by
// half a K-chunk period:
//
//      step:      0        1        2        3        4   ...
//      group 0:  stage0   mfma0    stage1   mfma1    stage2
//      group 1:   -       stage0   mfma0    stage1   mfma1
//
// so that on every SIMD one wave is always in its MFMA segment while its partner converts / stages the next
// chunk (global fp32 -> hi/lo f16 -> LDS, weights by LDS-DMA).  Two co-resident 256-thread blocks running the same
// program fall into lock step (both stage, then both fight for the matrix pipe: measured 0 overlap); the fixed
// anti-phase removes that (MI355X_MICROARCH.md "Two waves per SIMD", item 9).  One s_barrier per step.
//
// Epilogue: accumulators -> LDS [channel][256 pixels] -> every lane finishes 4 horizontally adjacent pixels and
// issues ONE 16-byte store (a wave writes eight 128-B row segments of one channel) instead of 64 dword stores;
// the optional style-gradient dot product is reduced per channel with one wave_sum.
#include "conv_common.hpp"
#include <cstdint>
#include <cstdlib>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int CK = 16, REC = 80;
constexpr int TR = 8, NT = 2, IN_R = 10, IN_C = 40, NPOS = IN_R * IN_C;
constexpr int OP = 260;                       // epilogue staging pitch (floats) per channel row

template <int MT>
constexpr int group_bytes() {
    constexpr int stage = NPOS * REC + 36 * 32 * MT * 16;
    constexpr int epi = 32 * MT * OP * 4;
    return (stage > epi ? stage : epi);
}

template <int MT, bool VEC>
__global__ __launch_bounds__(512) void conv_f16s_s1pp_kernel(const KArgs p, const uint4* __restrict__ wpk16, int total_items) {
    constexpr int MB = 32 * MT;
    constexpr int XBYTES = NPOS * REC;
    constexpr int WROWS = 36;
    constexpr int WPIECES = WROWS * MB * 16 / 1024;
    constexpr int NF4 = IN_R * (IN_C / 4);
    constexpr int XITEMS = VEC ? NF4 * 4 : NPOS * 4;
    constexpr int XPT = (XITEMS + 255) / 256;
    constexpr int XV = VEC ? 16 : 4;
    constexpr int GB = group_bytes<MT>();

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const oodgan_conv_args& a = p.a;
    const int grp = threadIdx.x >> 8;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    unsigned char* lx = smem + grp * GB;
    unsigned char* lw = lx + XBYTES;

    // work item of this group: two neighbouring tiles of the same weight block
    const int wi = 2 * xcd_remap(blockIdx.x, gridDim.x) + grp;
    const bool active = wi < total_items;
    BlockCtx ctx;
    {
        // work order: m-block fastest, then tile, then batch — the blocks that share an x tile (all its m-blocks)
        // run at the same time on one XCD, so the tile is fetched from HBM once and re-read from L2 (measured:
        // x re-reads from beyond L2 were the bound, 2.8 TB/s), and the two groups of a block share it in L1.
        int w = active ? wi : 0;
        const int ntile = p.tiles_x * p.tiles_y;
        ctx.mblk = w % p.mblocks;
        w /= p.mblocks;
        ctx.tile = w % ntile;
        ctx.b = w / ntile;
        ctx.r0 = (ctx.tile / p.tiles_x) * TR;
        ctx.c0 = (ctx.tile % p.tiles_x) * 32;
        ctx.m0 = ctx.mblk * MB;
    }
    const int b = ctx.b, r0 = ctx.r0, c0 = ctx.c0, m0 = ctx.m0;
    const float* xb = a.x + (long)b * a.K * p.in_plane;
    const float* isc = a.in_scale ? a.in_scale + (long)b * a.in_scale_stride : nullptr;
    const float* ish = a.in_shift ? a.in_shift + (long)b * a.in_scale_stride : nullptr;
    const float in_mul = a.in_mul2 ? a.in_mul2[1] : 1.f;
    const int row_org = r0 - 1, col_org = c0 - 4;

    int xoff[XPT], xpos[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int e = tid + i * 256;
        const int u = e >> 2;
        int r, c;
        if (VEC) { r = u / (IN_C / 4); c = 4 * (u % (IN_C / 4)); }
        else { r = u / IN_C; c = u % IN_C; }
        const int gy = row_org + r, gx = col_org + c;
        const bool ok = active && (e < XITEMS) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        xoff[i] = ok ? gy * a.in_pitch + gx : -1;
        xpos[i] = r * IN_C + c;
    }
    float xr[XPT][XV];
    const int nchunk = (a.K + CK - 1) / CK;
    const long wchunk = (long)WROWS * p.Mp;

    auto load_x = [&](int t) {
        const int k0 = t * CK + 4 * (tid & 3);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j;
                const bool ld = xoff[i] >= 0 && k < a.K && !(p.ablate & 2) && !((p.ablate & 32) && i == 1) && !((p.ablate & 64) && (j & 1));
                const float sc = ((ld && isc) ? isc[k] : 1.f) * in_mul;
                const float sh = ((ld && ish) ? ish[k] : 0.f) * in_mul;
                if (VEC) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ld) {
                        v = *reinterpret_cast<const float4*>(xb + (long)k * p.in_plane + xoff[i]);
                        v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
                    }
                    xr[i][j * 4 + 0] = v.x; xr[i][j * 4 + 1] = v.y; xr[i][j * 4 + 2] = v.z; xr[i][j * 4 + 3] = v.w;
                } else {
                    float v = 0.f;
                    if (ld) v = xb[(long)k * p.in_plane + xoff[i]] * sc + sh;
                    xr[i][j] = v;
                }
            }
        }
    };
    auto store_x = [&]() {
        const int q = tid & 3;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            if (e >= XITEMS) continue;
#pragma unroll
            for (int px = 0; px < (VEC ? 4 : 1); ++px) {
                half4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = VEC ? xr[i][j * 4 + px] : xr[i][j];
                    const _Float16 h = (_Float16)v;
                    hi[j] = h;
                    lo[j] = (_Float16)(v - (float)h);
                }
                const int pos = xpos[i] + px;
                *reinterpret_cast<half4*>(lx + pos * REC + q * 8) = hi;
                *reinterpret_cast<half4*>(lx + pos * REC + 32 + q * 8) = lo;
            }
        }
    };
    auto dma_w = [&](int t) {
#pragma unroll
        for (int i = 0; i < (WPIECES + 3) / 4; ++i) {
            const int pc = wave + i * 4;
            if (pc < WPIECES) {
                const int u = pc * 64 + lane;
                const int row = u / MB, j = u % MB;
                const uint4* src = wpk16 + (long)t * wchunk + (long)row * p.Mp + m0 + j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (lds_void*)(lw + pc * 1024), 16, 0, 0);
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;

    const unsigned char* lwh = lw + (half * MB + l31) * 16;
    const unsigned char* lxh = lx + ((wave * NT) * IN_C + l31 + 3) * REC + half * 16;
#define XFRAG(posoff, lo_) (*reinterpret_cast<const half8*>(lxh + (posoff) * REC + (lo_) * 32))
#define WFRAG(tap, lo_, mt) (*reinterpret_cast<const half8*>(lwh + ((((tap) * 2 + (lo_)) * 2) * MB + (mt) * 32) * 16))
#define MFMA3(accv, ah, al, bh, bl)                                              \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accv, 0, 0, 0);

    if (active) load_x(0);
    const int nsteps = 2 * nchunk + 1;
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();
        const int s = step - grp;
        if (!active || s < 0 || s >= 2 * nchunk) continue;
        const int t = s >> 1;
        if ((s & 1) == 0) {
            // ---- stage segment: start the weight DMA, convert the prefetched x chunk while it flies, then drain the
            // DMA explicitly: a bare s_barrier does not wait for LDS-DMA (hipcc emits no vmcnt here)
            if (!(p.ablate & 4)) dma_w(t);
            if (!(p.ablate & 8)) store_x();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // ---- matrix segment: prefetch chunk t+1 (registers, lands during the MFMAs), 9 taps x MT*NT*3 MFMAs
            if (t + 1 < nchunk) load_x(t + 1);
            if (p.ablate & 1) continue;
            // fragments of tap tp+1 are fetched (ds_read_b128) while the 12 MFMAs of tap tp run
            half8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
#define LOADF(buf, tp_)                                                                               \
    {                                                                                                 \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                           \
            ah[buf][mt] = WFRAG(tp_, 0, mt);                                                          \
            al[buf][mt] = WFRAG(tp_, 1, mt);                                                          \
        }                                                                                             \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                           \
            bh[buf][nt] = XFRAG((nt + (tp_) / 3) * IN_C + (tp_) % 3, 0);                              \
            bl[buf][nt] = XFRAG((nt + (tp_) / 3) * IN_C + (tp_) % 3, 1);                              \
        }                                                                                             \
    }
            LOADF(0, 0)
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int cur = tp & 1;
                if (tp + 1 < 9) LOADF(cur ^ 1, tp + 1)
                __builtin_amdgcn_sched_barrier(0);     // keep the next tap's 8 reads ahead of this tap's MFMAs
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { MFMA3(acc[mt][nt], ah[cur][mt], al[cur][mt], bh[cur][nt], bl[cur][nt]); }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef LOADF
        }
    }
#undef XFRAG
#undef WFRAG
#undef MFMA3
    __syncthreads();   // both groups finished reading their LDS images
    if (p.ablate & 16) { if (acc[0][0][0] == 123.456f) a.y[0] = 1.f; return; }

    // ---------------------------------------------------------------- epilogue through LDS
    float* lo = reinterpret_cast<float*>(lx);
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
    if (active) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    lo[m * OP + (wave * NT + nt) * 32 + l31] = acc[mt][nt][r] * us;
                }
    }
    __syncthreads();
    if (!active) return;
    const float* osc = a.out_scale ? a.out_scale + (long)b * a.out_scale_stride : nullptr;
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const long HWo = (long)p.Hout * p.Wout;
    const float* nzp = a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * HWo : nullptr;
    const float* db = a.dotx ? a.dotx + (long)b * a.M * HWo : nullptr;
    float* yb = a.y + (long)b * a.M * p.out_plane;
    const int c4 = tid & 63;                 // 4-pixel unit inside the 8x32 tile
    const int prow = c4 >> 3, pcol = (c4 & 7) * 4;
    const int py = r0 + prow, px = c0 + pcol;
    const bool row_ok = py < p.Hout;
#pragma unroll 4
    for (int i = 0; i < MB / 4; ++i) {
        const int ml = (tid >> 6) + 4 * i;   // wave-uniform channel
        const int m = m0 + ml;
        float4 v = *reinterpret_cast<const float4*>(lo + ml * OP + c4 * 4);
        float vv[4] = {v.x, v.y, v.z, v.w};
        float dsum = 0.f;
        const bool m_ok = m < a.M;
        if (m_ok && row_ok) {
            const long pix = (long)py * p.Wout + px;
            const float sc = osc ? osc[m] : 1.f;
            const float bv = a.bias ? a.bias[m] : 0.f;
            const float sl = (a.act == OODGAN_ACT_PRELU) ? a.slope[m] : 0.f;
            float* yp = yb + (long)m * p.out_plane + (long)py * a.out_pitch + px;
            if (VEC && px + 3 < p.Wout) {
                if (db) {
                    const float4 d4 = *reinterpret_cast<const float4*>(db + (long)m * HWo + pix);
                    dsum = vv[0] * d4.x + vv[1] * d4.y + vv[2] * d4.z + vv[3] * d4.w;
                }
                float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nzp) n4 = *reinterpret_cast<const float4*>(nzp + pix);
                const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float o = vv[j] * sc + nw * nn[j] + bv;
                    if (a.act == OODGAN_ACT_LRELU) o = (o > 0.f ? o : 0.2f * o) * kSqrt2;
                    else if (a.act == OODGAN_ACT_PRELU) o = o > 0.f ? o : sl * o;
                    vv[j] = o;
                }
                *reinterpret_cast<float4*>(yp) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (px + j >= p.Wout) continue;
                    if (db) dsum += vv[j] * db[(long)m * HWo + pix + j];
                    float o = vv[j] * sc + (nzp ? nw * nzp[pix + j] : 0.f) + bv;
                    if (a.act == OODGAN_ACT_LRELU) o = (o > 0.f ? o : 0.2f * o) * kSqrt2;
                    else if (a.act == OODGAN_ACT_PRELU) o = o > 0.f ? o : sl * o;
                    yp[j] = o;
                }
            }
        }
        if (db) {
            dsum = wave_sum(dsum);
            if (lane == 0 && m_ok) a.dot_part[((long)b * a.M + m) * a.dot_nparts + ctx.tile] = dsum;
        }
    }
}

}  // namespace

namespace oodgan {

int launch_s1pp(const oodgan_conv_args& a, const void* wpk16, const float* unscale, hipStream_t st) {
    KArgs p;
    p.a = a;
    p.w_unscale = unscale;
#ifdef OODGAN_DEBUG_ABLATE      // profiling builds only: the ablation bits make the kernel skip work (wrong results)
    { static int abl = getenv("OODGAN_ABLATE") ? atoi(getenv("OODGAN_ABLATE")) : 0; p.ablate = abl; }
#else
    p.ablate = 0;
#endif
    p.Hn = a.Hin; p.Wn = a.Win; p.Hout = a.Hin; p.Wout = a.Win;
    if (p.a.in_pitch == 0) p.a.in_pitch = a.Win;
    if (p.a.out_pitch == 0) p.a.out_pitch = p.Wout;
    p.in_plane = (long)a.Hin * p.a.in_pitch;
    p.out_plane = (long)p.Hout * p.a.out_pitch;
    p.tiles_y = (p.Hn + TR - 1) / TR;
    p.tiles_x = (p.Wn + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    const bool mt2 = a.M > 32;
    const int MB = mt2 ? 64 : 32;
    p.mblocks = (a.M + MB - 1) / MB;
    if (a.dotx) {
        OODGAN_REQUIRE(a.dot_part != nullptr, "conv3x3: dotx without dot_part");
        OODGAN_REQUIRE(a.dot_nparts == p.tiles_x * p.tiles_y, "conv3x3 f16s S1: dot_nparts %d != %d", a.dot_nparts,
                       p.tiles_x * p.tiles_y);
    }
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    const int items = (int)total;
    dim3 grid((unsigned)((total + 1) / 2)), block(512);
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
    const bool vec = (p.a.in_pitch % 4 == 0) && (p.a.out_pitch % 4 == 0) && (a.Win % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) &&
                     (!a.dotx || (reinterpret_cast<uintptr_t>(a.dotx) & 15) == 0) &&
                     (!a.noise || (reinterpret_cast<uintptr_t>(a.noise) & 15) == 0);
#define OODGAN_LAUNCH(MT_, VEC_)                                                                                         \
    {                                                                                                                    \
        constexpr int sm = 2 * group_bytes<MT_>();                                                                       \
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1pp_kernel<MT_, VEC_>),  \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, sm), true);            \
        (void)once;                                                                                                      \
        hipLaunchKernelGGL((conv_f16s_s1pp_kernel<MT_, VEC_>), grid, block, sm, st, p, w16, items);                      \
    }
    if (mt2) { if (vec) OODGAN_LAUNCH(2, true) else OODGAN_LAUNCH(2, false) }
    else { if (vec) OODGAN_LAUNCH(1, true) else OODGAN_LAUNCH(1, false) }
#undef OODGAN_LAUNCH
    return check_launch("conv3x3_f16s_s1pp");
}

}  // namespace oodgan
