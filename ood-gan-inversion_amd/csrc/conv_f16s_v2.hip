// Split-f16 3x3 stride-1 conv reading its input in S-form (sform.hpp): the x tile arrives by LDS-DMA exactly
// like the weights — no staging registers, no conversion VALU, contiguous 2176-byte runs.  Same ping-pong
// schedule, MFMA loop and LDS-staged epilogue as conv_f16s_pp.hip; additionally the epilogue can emit the
// S-form of (activated output x next layer's style) so that the next conv finds its input ready.
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>
#include <cstdlib>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int CK = 16, REC = 80;
constexpr int TR = 8, NT = 2, IN_R = 10, IN_C = 34, NPOS = IN_R * IN_C;   // 340 positions
constexpr int XSLOTS = NPOS * 5;                                          // 16-byte LDS slots incl. the pad slot
constexpr int XPIECES = (XSLOTS + 63) / 64;                               // 27 one-KiB DMA pieces
constexpr int XBYTES = XPIECES * 1024;                                    // 27648
constexpr int OP = 260;

template <int MT>
constexpr int group_bytes() {
    constexpr int stage = XBYTES + 36 * 32 * MT * 16;
    constexpr int epi = 32 * MT * OP * 4;
    return (stage > epi ? stage : epi);
}

struct SConv {
    const uint4* xs;        // S-form input
    SDims xd;
    uint4* ys;              // S-form output or null
    SDims yd;
    const float* ys_scale;  // (B,M) scale applied to the S-form output (next layer's style), stride ys_scale_stride
    int ys_scale_stride;
};

// Epilogue shared by the S1 and S2 S-form kernels: accumulators -> LDS [channel][256 pixels] -> 16-byte stores of
// the fp32 NCHW output (fused out-scale / noise / bias / activation, optional style-gradient dot) and, optionally,
// the S-form of the activated output for the next conv.  All threads of the workgroup must call it.
template <int MT>
__device__ __forceinline__ void tile_epilogue(const KArgs& p, const SConv& sc, f32x16 (&acc)[MT][2], const BlockCtx& ctx,
                                              unsigned char* lx, bool active, int tid, int lane, int wave, int l31, int half) {
    constexpr int MB = 32 * MT;
    constexpr int NT = 2;
    const oodgan_conv_args& a = p.a;
    const int b = ctx.b, r0 = ctx.r0, c0 = ctx.c0, m0 = ctx.m0;
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
    float* yb = a.y ? a.y + (long)b * a.M * p.out_plane : nullptr;
    const bool vec = (p.Wout % 4 == 0) && (a.out_pitch % 4 == 0);
    if (vec) {
        // Aligned rows (every layer of the generator and the encoder): the constants, the noise quad and the dot operands of EIGHT
        // channels are requested together from clamped channels / pixels (`lo`-independent; an absent tensor reads the workgroup's
        // own output row, a valid address) and masked where they are used.  The loop below — a conditional load per constant and
        // operand, each one waited for with vmcnt(0) behind the previous channel's store — was 120 loads with 125 full waits:
        // 25-40 round trips per tile in kernels of 40-75 us.
        const int c4 = tid & 63;
        const int prow = c4 >> 3, pcol = (c4 & 7) * 4;
        const int py = r0 + prow, px = c0 + pcol;
        const bool pix_ok = py < p.Hout && px + 3 < p.Wout;
        const long pixc = (long)min(py, p.Hout - 1) * p.Wout + min(px, p.Wout - 4);
        const float* dummy = reinterpret_cast<const float*>(sc.xs);      // the S-form input: always there, 16-byte aligned
        const float4 n4 = *reinterpret_cast<const float4*>(nzp ? nzp + pixc : dummy);
        const float nn[4] = {nzp ? n4.x : 0.f, nzp ? n4.y : 0.f, nzp ? n4.z : 0.f, nzp ? n4.w : 0.f};
#pragma unroll
        for (int i0 = 0; i0 < MB / 4; i0 += 8) {
            float sclv[8], bvv[8], slv[8];
            float4 d4v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int mc = min(m0 + (tid >> 6) + 4 * (i0 + u), a.M - 1);
                sclv[u] = *(osc ? osc + mc : dummy);
                bvv[u] = *(a.bias ? a.bias + mc : dummy);
                slv[u] = *((a.act == OODGAN_ACT_PRELU && a.slope) ? a.slope + mc : dummy);
                d4v[u] = *reinterpret_cast<const float4*>(db ? db + (long)mc * HWo + pixc : dummy);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ml = (tid >> 6) + 4 * (i0 + u);
                const int m = m0 + ml;
                const float4 v = *reinterpret_cast<const float4*>(lo + ml * OP + c4 * 4);
                float vv[4] = {v.x, v.y, v.z, v.w};
                float dsum = 0.f;
                const bool m_ok = m < a.M;
                if (m_ok && pix_ok && (yb || db)) {
                    const float scl = osc ? sclv[u] : 1.f, bv = a.bias ? bvv[u] : 0.f, sl = slv[u];
                    if (db) dsum = vv[0] * d4v[u].x + vv[1] * d4v[u].y + vv[2] * d4v[u].z + vv[3] * d4v[u].w;
                    if (yb) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float o = vv[j] * scl + nw * nn[j] + bv;
                            if (a.act == OODGAN_ACT_LRELU) o = (o > 0.f ? o : 0.2f * o) * kSqrt2;
                            else if (a.act == OODGAN_ACT_PRELU) o = o > 0.f ? o : sl * o;
                            vv[j] = o;
                        }
                        *reinterpret_cast<float4*>(yb + (long)m * p.out_plane + (long)py * a.out_pitch + px) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    }
                }
                if (db) {
                    dsum = wave_sum(dsum);
                    if (lane == 0 && m_ok) a.dot_part[((long)b * a.M + m) * a.dot_nparts + ctx.tile] = dsum;
                }
            }
        }
    } else {
        const int c4 = tid & 63;
        const int prow = c4 >> 3, pcol = (c4 & 7) * 4;
        const int py = r0 + prow, px = c0 + pcol;
        const bool row_ok = py < p.Hout;
#pragma unroll 4
        for (int i = 0; i < MB / 4; ++i) {
            const int ml = (tid >> 6) + 4 * i;
            const int m = m0 + ml;
            float4 v = *reinterpret_cast<const float4*>(lo + ml * OP + c4 * 4);
            float vv[4] = {v.x, v.y, v.z, v.w};
            float dsum = 0.f;
            const bool m_ok = m < a.M;
            if (m_ok && row_ok && (yb || db)) {
                const long pix = (long)py * p.Wout + px;
                const float scl = osc ? osc[m] : 1.f;
                const float bv = a.bias ? a.bias[m] : 0.f;
                const float sl = (a.act == OODGAN_ACT_PRELU) ? a.slope[m] : 0.f;
                float* yp = yb ? yb + (long)m * p.out_plane + (long)py * a.out_pitch + px : nullptr;
                if (vec && px + 3 < p.Wout) {
                    if (db) {
                        const float4 d4 = *reinterpret_cast<const float4*>(db + (long)m * HWo + pix);
                        dsum = vv[0] * d4.x + vv[1] * d4.y + vv[2] * d4.z + vv[3] * d4.w;
                    }
                    if (yp) {
                        float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (nzp) n4 = *reinterpret_cast<const float4*>(nzp + pix);
                        const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float o = vv[j] * scl + nw * nn[j] + bv;
                            if (a.act == OODGAN_ACT_LRELU) o = (o > 0.f ? o : 0.2f * o) * kSqrt2;
                            else if (a.act == OODGAN_ACT_PRELU) o = o > 0.f ? o : sl * o;
                            vv[j] = o;
                        }
                        *reinterpret_cast<float4*>(yp) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (px + j >= p.Wout) continue;
                        if (db) dsum += vv[j] * db[(long)m * HWo + pix + j];
                        if (yp) {
                            float o = vv[j] * scl + (nzp ? nw * nzp[pix + j] : 0.f) + bv;
                            if (a.act == OODGAN_ACT_LRELU) o = (o > 0.f ? o : 0.2f * o) * kSqrt2;
                            else if (a.act == OODGAN_ACT_PRELU) o = o > 0.f ? o : sl * o;
                            yp[j] = o;
                        }
                    }
                }
            }
            if (db) {
                dsum = wave_sum(dsum);
                if (lane == 0 && m_ok) a.dot_part[((long)b * a.M + m) * a.dot_nparts + ctx.tile] = dsum;
            }
        }
    }
    // ---- optional S-form output: (activated output) * ys_scale[b,m], split hi/lo, 16 channels per 64-byte record
    if (sc.ys) {
        const float* ysc = sc.ys_scale ? sc.ys_scale + (long)b * sc.ys_scale_stride : nullptr;
        // unit = (pixel of the 8x32 tile, 16-channel block): the thread activates 16 channels once and writes the
        // whole 64-byte record (4 x 16 B, consecutive lanes -> consecutive records: 4 KB contiguous per wave)
        for (int u = tid; u < 256 * (MB / 16); u += 256) {
            const int pxl = u & 255, g16 = u >> 8;
            const int py = r0 + (pxl >> 5), px = c0 + (pxl & 31);
            if (py >= p.Hout || px >= p.Wout || (m0 >> 4) + g16 >= sc.yd.KC) continue;   // last M block may be partial
            // the constants of the unit's 16 channels in one request (clamped channel; the S-form input as a valid address for an absent
            // tensor): loaded where they are used they are up to 64 conditional loads, each waited for with vmcnt(0)
            const float* dmy = reinterpret_cast<const float*>(sc.xs);
            const float nzr = *(nzp ? nzp + (long)py * p.Wout + px : dmy);
            float c_osc[16], c_bia[16], c_slp[16], c_ysc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int mc = min(m0 + g16 * 16 + j, a.M - 1);
                c_osc[j] = *(osc ? osc + mc : dmy);
                c_bia[j] = *(a.bias ? a.bias + mc : dmy);
                c_slp[j] = *((a.act == OODGAN_ACT_PRELU && a.slope) ? a.slope + mc : dmy);
                c_ysc[j] = *(ysc ? ysc + mc : dmy);
            }
            const float nz = nzp ? nw * nzr : 0.f;
            half8 hv[2], lv[2];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ml = g16 * 16 + j, m = m0 + ml;
                float o = 0.f;
                if (m < a.M) {
                    o = lo[ml * OP + pxl] * (osc ? c_osc[j] : 1.f) + nz + (a.bias ? c_bia[j] : 0.f);
                    if (a.act == OODGAN_ACT_LRELU) o = (o > 0.f ? o : 0.2f * o) * kSqrt2;
                    else if (a.act == OODGAN_ACT_PRELU) o = o > 0.f ? o : c_slp[j] * o;
                    if (ysc) o *= c_ysc[j];
                }
                const _Float16 h = (_Float16)o;
                hv[j >> 3][j & 7] = h;
                lv[j >> 3][j & 7] = (_Float16)(o - (float)h);
            }
            half8* rec = reinterpret_cast<half8*>(sc.ys + sform_unit(sc.yd, b, (m0 >> 4) + g16, py, px, 0));
            rec[0] = hv[0]; rec[1] = hv[1]; rec[2] = lv[0]; rec[3] = lv[1];
        }
    }
}

// CPS = K chunks fetched per stage.  Layers with few tiles (4x4 ... 16x16 images) put at most one workgroup on a CU, so
// every stage costs a full DMA latency; they run as NG = 1, CPS = 2: half the stages, twice the bytes in flight.
template <int MT, int NG, int CPS = 1>
__global__ __launch_bounds__(256 * NG) void conv_f16s_s1v2_kernel(const KArgs p, const uint4* __restrict__ wpk16, int total_items,
                                                             const SConv sc) {
    constexpr int MB = 32 * MT;
    constexpr int WROWS = 36;
    constexpr int WPIECES = WROWS * MB * 16 / 1024;
    constexpr int STAGE = XBYTES + WROWS * MB * 16;
    constexpr int GB = (STAGE * CPS > group_bytes<MT>()) ? STAGE * CPS : group_bytes<MT>();
    constexpr int XPW = (XPIECES + 3) / 4;          // x pieces per wave

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const oodgan_conv_args& a = p.a;
    const int grp = threadIdx.x >> 8;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    unsigned char* lx = smem + grp * GB;
    unsigned char* lw = lx + XBYTES;

    const int wi = NG * xcd_remap(blockIdx.x, gridDim.x) + grp;
    const bool active = wi < total_items;
    BlockCtx ctx;
    {
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

    // per-lane DMA source offsets (16-byte units inside one (b,kc) plane) of this wave's x pieces
    long xsrc[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int pc = wave + 4 * i;
        int P = pc * 64 + lane;                 // physical LDS slot
        if (P >= XSLOTS) P = XSLOTS - 1;        // tail of the last piece: harmless duplicate
        const int pos = P / 5;
        int s = P % 5;
        if (s == 4) s = 0;                      // pad slot: any valid address
        const int r = pos / IN_C, c = pos % IN_C;
        // tile origin = image (r0-1, c0-1) = padded (r0, c0)
        xsrc[i] = ((long)(r0 + r) * sc.xd.Wp + (c0 + c)) * 4 + s;
    }
    const uint4* xplane0 = sc.xs + (long)b * sc.xd.KC * sc.xd.plane;
    const int nchunk = (a.K + CK - 1) / CK;
    const long wchunk = (long)WROWS * p.Mp;

    auto dma_x = [&](int t, int sub) {
        const uint4* base = xplane0 + (long)t * sc.xd.plane;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int pc = wave + 4 * i;
            if (pc < XPIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xsrc[i]),
                                                 (lds_void*)(lx + sub * STAGE + pc * 1024), 16, 0, 0);
        }
    };
    auto dma_w = [&](int t, int sub) {
#pragma unroll
        for (int i = 0; i < (WPIECES + 3) / 4; ++i) {
            const int pc = wave + i * 4;
            if (pc < WPIECES) {
                const int u = pc * 64 + lane;
                const int row = u / MB, j = u % MB;
                const uint4* src = wpk16 + (long)t * wchunk + (long)row * p.Mp + m0 + j;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (lds_void*)(lw + sub * STAGE + pc * 1024), 16, 0, 0);
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
    const unsigned char* lxh = lx + ((wave * NT) * IN_C + l31) * REC + half * 16;
#define MFMA3(accv, ah, al, bh, bl)                                              \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accv, 0, 0, 0);

    const int nstage = (nchunk + CPS - 1) / CPS;
    const int nsteps = 2 * nstage + 1;
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();
        const int s = step - grp;
        if (!active || s < 0 || s >= 2 * nstage) continue;
        const int t = s >> 1;
        if ((s & 1) == 0) {
            // stage segment: both operands by LDS-DMA; drained explicitly (a bare s_barrier does not wait for it)
#pragma unroll
            for (int sub = 0; sub < CPS; ++sub)
                if (t * CPS + sub < nchunk) {
                    dma_x(t * CPS + sub, sub);
                    dma_w(t * CPS + sub, sub);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int sub = 0; sub < CPS; ++sub) {
            if (t * CPS + sub >= nchunk) break;
            const unsigned char* lxs = lxh + sub * STAGE;
            const unsigned char* lws = lwh + sub * STAGE;
#define XFRAG(posoff, lo_) (*reinterpret_cast<const half8*>(lxs + (posoff) * REC + (lo_) * 32))
#define WFRAG(tap, lo_, mt) (*reinterpret_cast<const half8*>(lws + ((((tap) * 2 + (lo_)) * 2) * MB + (mt) * 32) * 16))
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
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { MFMA3(acc[mt][nt], ah[cur][mt], al[cur][mt], bh[cur][nt], bl[cur][nt]); }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef LOADF
#undef XFRAG
#undef WFRAG
            }
        }
    }
#undef MFMA3
    __syncthreads();

    tile_epilogue<MT>(p, sc, acc, ctx, lx, active, tid, lane, wave, l31, half);
}

// Low-resolution instance (4x4 ... 32x32 images: at most 256 work items, one workgroup per CU, K = 512): the <1,1,2>
// instance above requests a stage, waits for it (vmcnt(0)) and only then issues its MFMAs — 16 stages of a bare LDS-DMA
// latency (3.2 us for 92 KB) plus their MFMAs (1.5 us), nothing overlapped.  Here the stages form a RING of three 46 KB
// buffers (one K chunk each): stage t+2 is requested right after the barrier that retires stage t-1, and the wait before
// stage t leaves stage t+1 in flight (counted vmcnt: every wave issues exactly RING_NPW pieces per stage, the tail pieces
// duplicated).  One barrier per stage.
constexpr int RING = 3;
template <int MT>
__global__ __launch_bounds__(256) void conv_f16s_s1ring_kernel(const KArgs p, const uint4* __restrict__ wpk16, int total_items,
                                                                const SConv sc) {
    constexpr int MB = 32 * MT;
    constexpr int WROWS = 36;
    constexpr int WPIECES = WROWS * MB * 16 / 1024;
    constexpr int STAGE = XBYTES + WROWS * MB * 16;
    constexpr int NPIECES = XPIECES + WPIECES;
    constexpr int NPW = (NPIECES + 3) / 4;          // pieces per wave and stage (uniform: counted waits)
    static_assert(RING * STAGE >= group_bytes<MT>(), "the epilogue image fits the ring");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    const int wi = xcd_remap(blockIdx.x, gridDim.x);
    const bool active = wi < total_items;
    BlockCtx ctx;
    {
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

    // per-lane source offsets (16-byte units) of this wave's pieces: x pieces inside one (b,kc) plane, weight pieces inside
    // one K chunk of the packed weights
    long src[NPW];
    bool isx[NPW];
    int pcs[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        int pc = wave + 4 * i;
        if (pc >= NPIECES) pc = NPIECES - 1;    // duplicate of the last piece: same bytes to the same place
        pcs[i] = pc;
        isx[i] = pc < XPIECES;
        if (pc < XPIECES) {
            int P = pc * 64 + lane;
            if (P >= XSLOTS) P = XSLOTS - 1;
            const int pos = P / 5;
            int sl = P % 5;
            if (sl == 4) sl = 0;
            const int r = pos / IN_C, c = pos % IN_C;
            src[i] = ((long)(r0 + r) * sc.xd.Wp + (c0 + c)) * 4 + sl;
        } else {
            const int u = (pc - XPIECES) * 64 + lane;
            src[i] = (long)(u / MB) * p.Mp + m0 + (u % MB);
        }
    }
    const uint4* xplane0 = sc.xs + (long)b * sc.xd.KC * sc.xd.plane;
    const int nchunk = (a.K + CK - 1) / CK;
    const long wchunk = (long)WROWS * p.Mp;
    auto dma_stage = [&](int t) {
        unsigned char* dst = smem + (t % RING) * STAGE;
        const uint4* xb = xplane0 + (long)t * sc.xd.plane;
        const uint4* wb = wpk16 + (long)t * wchunk;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const uint4* sp = (isx[i] ? xb : wb) + src[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp, (lds_void*)(dst + pcs[i] * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;

    const unsigned lwoff = XBYTES + (half * MB + l31) * 16;
    const unsigned lxoff = ((wave * NT) * IN_C + l31) * REC + half * 16;
#define MFMA3(accv, ah, al, bh, bl)                                              \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accv, 0, 0, 0);
    if (active) {
        dma_stage(0);
        if (nchunk > 1) dma_stage(1);
    }
    for (int t = 0; t < nchunk; ++t) {
        if (t + 1 < nchunk) __builtin_amdgcn_s_waitcnt(((NPW >> 4) & 3) << 14 | 0x0F70 | (NPW & 15));       // stage t+1 stays in flight
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();                  // stage t has landed for every wave; everybody is done with stage t-1
        if (!active) continue;
        if (t + 2 < nchunk) dma_stage(t + 2);          // into the buffer stage t-1 occupied
        const unsigned char* lxs = smem + (t % RING) * STAGE + lxoff;
        const unsigned char* lws = smem + (t % RING) * STAGE + lwoff;
#define XFRAG(posoff, lo_) (*reinterpret_cast<const half8*>(lxs + (posoff) * REC + (lo_) * 32))
#define WFRAG(tap, lo_, mt) (*reinterpret_cast<const half8*>(lws + ((((tap) * 2 + (lo_)) * 2) * MB + (mt) * 32) * 16))
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
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { MFMA3(acc[mt][nt], ah[cur][mt], al[cur][mt], bh[cur][nt], bl[cur][nt]); }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef LOADF
#undef XFRAG
#undef WFRAG
    }
#undef MFMA3
    __syncthreads();
    tile_epilogue<MT>(p, sc, acc, ctx, smem, active, tid, lane, wave, l31, half);
}

// ---------------------------------------------------------------------------------------------------------------
// Transposed 3x3 stride-2 conv (the up-sampling ModulatedConv2d before its blur) with S-form input: block = 4 waves,
// N tile = 4 rows x 32 positions (i',j') of the (H+1)x(W+1) position grid, 4 output phases per position.
// x tile = rows r0-1..r0+3, cols c0-1..c0+31 of the S-form image (165 records, 13 one-KiB DMA pieces).
constexpr int T2_R = 5, T2_C = 33, T2_NPOS = T2_R * T2_C;
constexpr int T2_XPIECES = (T2_NPOS * 5 + 63) / 64;          // 13
constexpr int T2_XBYTES = T2_XPIECES * 1024;

template <int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MT == 1 ? 4 : 1, MT == 1 ? 4 : 1))) void conv_f16s_t2v2_kernel(const KArgs p, const uint4* __restrict__ wpk16, const SConv sc) {
    constexpr int MB = 32 * MT;
    constexpr int WROWS = 36;
    constexpr int WPIECES = WROWS * MB * 16 / 1024;
    constexpr int XPW = (T2_XPIECES + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lx = smem;
    unsigned char* lw = smem + T2_XBYTES;
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    BlockCtx ctx;
    {
        int w = xcd_remap(blockIdx.x, gridDim.x);
        const int ntile = p.tiles_x * p.tiles_y;
        ctx.mblk = w % p.mblocks;
        w /= p.mblocks;
        ctx.tile = w % ntile;
        ctx.b = w / ntile;
        ctx.r0 = (ctx.tile / p.tiles_x) * 4;
        ctx.c0 = (ctx.tile % p.tiles_x) * 32;
        ctx.m0 = ctx.mblk * MB;
    }
    const int b = ctx.b, r0 = ctx.r0, c0 = ctx.c0, m0 = ctx.m0;
    long xsrc[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int pc = wave + 4 * i;
        int P = pc * 64 + lane;
        if (P >= T2_NPOS * 5) P = T2_NPOS * 5 - 1;
        const int pos = P / 5;
        int s = P % 5;
        if (s == 4) s = 0;
        const int r = pos / T2_C, c = pos % T2_C;
        xsrc[i] = ((long)(r0 + r) * sc.xd.Wp + (c0 + c)) * 4 + s;      // tile origin (r0-1,c0-1) = padded (r0,c0)
    }
    const uint4* xplane0 = sc.xs + (long)b * sc.xd.KC * sc.xd.plane;
    const int nchunk = (a.K + CK - 1) / CK;
    const long wchunk = (long)WROWS * p.Mp;

    f32x16 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;
    const unsigned char* lwh = lw + (half * MB + l31) * 16;
    const unsigned char* lxh = lx + (wave * T2_C + l31) * REC + half * 16;
#define XFRAG(posoff, lo_) (*reinterpret_cast<const half8*>(lxh + (posoff) * REC + (lo_) * 32))
#define WFRAG(tap, lo_, mt) (*reinterpret_cast<const half8*>(lwh + ((((tap) * 2 + (lo_)) * 2) * MB + (mt) * 32) * 16))
#define MFMA3(accv, ah, al, bh, bl)                                              \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accv, 0, 0, 0);
    for (int t = 0; t < nchunk; ++t) {
        __syncthreads();
        {
            const uint4* base = xplane0 + (long)t * sc.xd.plane;
#pragma unroll
            for (int i = 0; i < XPW; ++i) {
                const int pc = wave + 4 * i;
                if (pc < T2_XPIECES)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xsrc[i]),
                                                     (lds_void*)(lx + pc * 1024), 16, 0, 0);
            }
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        half8 bh[2][2], bl[2][2];
#pragma unroll
        for (int da = 0; da < 2; ++da)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                bh[da][db] = XFRAG((1 - da) * T2_C + (1 - db), 0);
                bl[da][db] = XFRAG((1 - da) * T2_C + (1 - db), 1);
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int ky = tp / 3, kx = tp % 3;
                const int ph = (ky & 1) * 2 + (kx & 1);
                const half8 ah = WFRAG(tp, 0, mt), al = WFRAG(tp, 1, mt);
                MFMA3(acc[mt][ph], ah, al, bh[ky >> 1][kx >> 1], bl[ky >> 1][kx >> 1]);
            }
        }
    }
#undef XFRAG
#undef WFRAG
#undef MFMA3
    conv_epilogue<OODGAN_CONV_T2, MT, 1, 4>(p, acc, ctx, wave, l31, half);
}


// ---------------------------------------------------------------------------------------------------------------
// 3x3 stride-2 conv (input gradient of the transposed up-conv) on a PHASE-SPLIT S-form input:
//   dx[m,i,j] = sum_{k,ky,kx} g2[k, 2i+ky, 2j+kx] W[ky,kx]  =  sum over the four parity images
//   G_{py,px}[i,j] = g2[2i+py, 2j+px]  of stride-1 taps  G_{py,px}[i+a, j+b] W[2a+py, 2b+px]   (a <= (2-py)/2, ...)
// so every tile row is again one contiguous run.  SP[b][kc][ph=py*2+px][Hq][Wq][4 slots].
// K stage = (16-channel chunk, row parity py): x tile = 9 rows x {px=0,1} x 33 cols (594 records), taps ky = py, py+2.
// Two anti-phase groups as in the S1 kernel; same LDS epilogue.
constexpr int S2_ROWS = 9, S2_C = 33, S2_NPOS = S2_ROWS * 2 * S2_C;          // 594
constexpr int S2_XPIECES = (S2_NPOS * 5 + 63) / 64;                          // 47
constexpr int S2_XBYTES = S2_XPIECES * 1024;

template <int MT>
constexpr int s2_group_bytes() {
    constexpr int stage = S2_XBYTES + 6 * 4 * 32 * MT * 16;
    constexpr int epi = 32 * MT * OP * 4;
    return (stage > epi ? stage : epi);
}

struct SPDims { int KC, Hq, Wq; long plane; };   // plane = Hq*Wq*4 (16-byte units) per (b,kc,phase)

__host__ __device__ inline SPDims sp_dims(int C, int H, int W) {    // H,W = S2 OUTPUT size
    SPDims d;
    d.KC = (C + 15) / 16;
    d.Hq = (H + 7) / 8 * 8 + 2;
    d.Wq = (W + 31) / 32 * 32 + 2;
    d.plane = (long)d.Hq * d.Wq * 4;
    return d;
}

// PYM: both row parities (all 9 taps) of a channel chunk in ONE stage — half the stages for layers with so few tiles
// that every stage is a bare DMA latency (single group: the merged stage needs 112 KB of LDS at MT = 1).
template <int MT, int NG = 2, bool PYM = false>
__global__ __launch_bounds__(256 * NG) void conv_f16s_s2v2_kernel(const KArgs p, const uint4* __restrict__ wpk16, int total_items,
                                                             const SConv sc, const SPDims sp) {
    constexpr int MB = 32 * MT;
    constexpr int PYMB = 2 * S2_XBYTES + 9 * 4 * MB * 16;
    constexpr int GB = PYM ? (PYMB > s2_group_bytes<MT>() ? PYMB : s2_group_bytes<MT>()) : s2_group_bytes<MT>();
    constexpr int XPW = (S2_XPIECES + 3) / 4;       // 12
    constexpr int TPP = 4 * MB * 16 / 1024;         // DMA pieces per tap (hi|lo x h x MB x 16 B)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const oodgan_conv_args& a = p.a;
    const int grp = threadIdx.x >> 8;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    unsigned char* lx = smem + grp * GB;
    unsigned char* lw = lx + (PYM ? 2 : 1) * S2_XBYTES;
    const int wi = NG * xcd_remap(blockIdx.x, gridDim.x) + grp;
    const bool active = wi < total_items;
    BlockCtx ctx;
    {
        int w = active ? wi : 0;
        const int ntile = p.tiles_x * p.tiles_y;
        ctx.mblk = w % p.mblocks;
        w /= p.mblocks;
        ctx.tile = w % ntile;
        ctx.b = w / ntile;
        ctx.r0 = (ctx.tile / p.tiles_x) * 8;
        ctx.c0 = (ctx.tile % p.tiles_x) * 32;
        ctx.m0 = ctx.mblk * MB;
    }
    const int b = ctx.b, r0 = ctx.r0, c0 = ctx.c0, m0 = ctx.m0;
    // per-lane DMA source offsets for py = 0 (add 2*plane for py = 1)
    long xsrc[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int pc = wave + 4 * i;
        int P = pc * 64 + lane;
        if (P >= S2_NPOS * 5) P = S2_NPOS * 5 - 1;
        const int pos = P / 5;
        int s = P % 5;
        if (s == 4) s = 0;
        const int rr = pos / (2 * S2_C), px = (pos % (2 * S2_C)) / S2_C, cc = pos % S2_C;
        xsrc[i] = (long)px * sp.plane + ((long)(r0 + rr) * sp.Wq + (c0 + cc)) * 4 + s;
    }
    const uint4* xb0 = sc.xs + (long)b * sp.KC * 4 * sp.plane;
    const int nchunk = (a.K + CK - 1) / CK;
    const long wchunk = (long)36 * p.Mp;

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;
    const unsigned char* lwh = lw + (half * MB + l31) * 16;
    const unsigned char* lxh = lx + ((wave * 2) * 2 * S2_C + l31) * REC + half * 16;
#define XFRAGB(buf, posoff, lo_) (*reinterpret_cast<const half8*>(lxh + (buf) * S2_XBYTES + (posoff) * REC + (lo_) * 32))
#define WSLOT(slot, lo_, mt) (*reinterpret_cast<const half8*>(lwh + ((((slot) * 2 + (lo_)) * 2) * MB + (mt) * 32) * 16))
#define MFMA3(accv, ah, al, bh, bl)                                              \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accv, 0, 0, 0);
    // taps (a, kx): LDS slot; x position: row nt + a, plane px = kx&1, col shift kx>>1, x buffer buf
#define S2_TAP(slot_, a_, kx_, buf_)                                                                             \
    {                                                                                                            \
        half8 ahv[MT], alv[MT];                                                                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { ahv[mt] = WSLOT(slot_, 0, mt); alv[mt] = WSLOT(slot_, 1, mt); } \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                                       \
            const half8 bhv = XFRAGB(buf_, ((nt + (a_)) * 2 + ((kx_) & 1)) * S2_C + ((kx_) >> 1), 0);            \
            const half8 blv = XFRAGB(buf_, ((nt + (a_)) * 2 + ((kx_) & 1)) * S2_C + ((kx_) >> 1), 1);            \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { MFMA3(acc[mt][nt], ahv[mt], alv[mt], bhv, blv); } \
        }                                                                                                        \
    }

    auto dma_x = [&](int t, int py, int buf) {
        const uint4* base = xb0 + ((long)t * 4 + py * 2) * sp.plane;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int pc = wave + 4 * i;
            if (pc < S2_XPIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xsrc[i]),
                                                 (lds_void*)(lx + buf * S2_XBYTES + pc * 1024), 16, 0, 0);
        }
    };
    // weights: LDS slot order = py-0 taps (ky in {0,2}: slot (ky>>1)*3 + kx, 6 taps), then (PYM) py-1 taps (ky = 1, slots 6..8)
    auto dma_w = [&](int t, int slot0, int nslot, bool py1) {
        for (int pc = wave; pc < nslot * TPP; pc += 4) {
            const int sl = pc / TPP;
            const int tap = py1 ? (3 + sl) : ((sl / 3) * 6 + sl % 3);
            const int u = (pc % TPP) * 64 + lane;                 // 16-byte unit inside the tap block [hl][h][MB]
            const int row = u / MB, j = u % MB;
            const uint4* src = wpk16 + (long)t * wchunk + (long)(tap * 4 + row) * p.Mp + m0 + j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (lds_void*)(lw + (slot0 * TPP + pc) * 1024), 16, 0, 0);
        }
    };

    const int nstage = PYM ? nchunk : 2 * nchunk;  // (chunk) or (chunk, py)
    const int nsteps = 2 * nstage + (NG - 1);
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();
        const int sidx = step - grp;
        if (!active || sidx < 0 || sidx >= 2 * nstage) continue;
        const int q = sidx >> 1;
        if constexpr (PYM) {
            if ((sidx & 1) == 0) {
                dma_x(q, 0, 0);
                dma_x(q, 1, 1);
                dma_w(q, 0, 6, false);
                dma_w(q, 6, 3, true);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                S2_TAP(0, 0, 0, 0) S2_TAP(1, 0, 1, 0) S2_TAP(2, 0, 2, 0)
                S2_TAP(3, 1, 0, 0) S2_TAP(4, 1, 1, 0) S2_TAP(5, 1, 2, 0)
                S2_TAP(6, 0, 0, 1) S2_TAP(7, 0, 1, 1) S2_TAP(8, 0, 2, 1)
            }
        } else {
            const int t = q >> 1, py = q & 1;
            if ((sidx & 1) == 0) {
                dma_x(t, py, 0);
                dma_w(t, 0, py ? 3 : 6, py != 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                S2_TAP(0, 0, 0, 0) S2_TAP(1, 0, 1, 0) S2_TAP(2, 0, 2, 0)
                if (py == 0) { S2_TAP(3, 1, 0, 0) S2_TAP(4, 1, 1, 0) S2_TAP(5, 1, 2, 0) }
            }
        }
    }
#undef S2_TAP
#undef XFRAGB
#undef WSLOT
#undef MFMA3
    __syncthreads();
    tile_epilogue<MT>(p, sc, acc, ctx, lx, active, tid, lane, wave, l31, half);
}

// pitched fp32 g2 (B,C,2H+1,pitch) -> phase-split S-form of g2*scale[b,c]*mul2[1]; thread = one 64-byte record
__global__ __launch_bounds__(256) void to_sform_phases_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              int scale_stride, const float* __restrict__ mul2,
                                                              uint4* __restrict__ out, int B, int C, int H, int W, int Hin, int Win,
                                                              int in_pitch, SPDims sp, int pad) {
    // pad = 1: x is the UNPADDED (B,C,2H,2W-ish) tensor of a conv with padding 1 — source pixel (yy-1, xx-1), row / column 0 of the
    // padded image are zeros (the zero-pad copy the forward use of the stride-2 conv used to make with two torch kernels)
    const int Hh = H + 1, Wh = W + 1;
    const long total = (long)B * sp.KC * 4 * Hh * Wh;
    const float gm = mul2 ? mul2[1] : 1.f;
    const long in_plane = (long)Hin * in_pitch;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int j = (int)(e % Wh);
        const int i = (int)((e / Wh) % Hh);
        const int ph = (int)((e / ((long)Wh * Hh)) % 4);
        const int kc = (int)((e / ((long)Wh * Hh * 4)) % sp.KC);
        const int b = (int)(e / ((long)Wh * Hh * 4 * sp.KC));
        const int yy = 2 * i + (ph >> 1) - pad, xx = 2 * j + (ph & 1) - pad;
        const bool inb = yy >= 0 && xx >= 0 && yy < Hin && xx < Win;
        half8 h0, h1, l0, l1;
        // one request for the record's 16 channels and scales (clamped pixel and channel), masked below — see to_sform_kernel
        float xv[16], scv[16];
        const float* scp = scale ? scale + (long)b * scale_stride : x;
        const long pofs = (long)max(0, min(yy, Hin - 1)) * in_pitch + max(0, min(xx, Win - 1));
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int cc = min(kc * 16 + jj, C - 1);
            xv[jj] = x[((long)b * C + cc) * in_plane + pofs];
            scv[jj] = scp[cc];
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int c = kc * 16 + jj;
            float v = 0.f;
            if (inb && c < C) v = xv[jj] * (scale ? scv[jj] : 1.f) * gm;
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            if (jj < 8) { h0[jj] = h; l0[jj] = l; } else { h1[jj - 8] = h; l1[jj - 8] = l; }
        }
        half8* o = reinterpret_cast<half8*>(out + ((((long)b * sp.KC + kc) * 4 + ph) * sp.plane + ((long)i * sp.Wq + j) * 4));
        o[0] = h0; o[1] = h1; o[2] = l0; o[3] = l1;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// blur^T + phase split + f16 split in one pass: the producer of the S2 conv's input.
//   g2 = upfirdn2d(g, flip(k), pad=(2,2))  ((2H)x(2W) -> (2H+1)x(2W+1), the adjoint of Blur(pad=(1,1)), reference
//   src/ops/op/upfirdn2d.py:115-120), times scale[b,c]*mul2[1], written as SP[b][kc][py*2+px][i][j] records.
// Block = one (b, 16-channel block) x a tile of 4x32 (i,j) positions = g2 rows 2*i0..2*i0+7, cols 2*j0..2*j0+63.
//   A  load g rows [2*i0-2, 2*i0+8], cols [2*j0-4, 2*j0+68) for 16 channels into LDS (float4, zero fill)
//   B  thread (channel, row, half-row) filters 32 horizontally adjacent outputs from a sliding register window
//   C  results are exchanged through LDS ([channel][position], reusing the input region) so that one thread owns
//      one position with all 16 channels, splits hi/lo and writes the 64-byte record (coalesced 4 KB per wave).
constexpr int BT_R = 11, BT_C = 72;            // input tile rows / cols (cols: global 2*j0-4 ..)

__global__ __launch_bounds__(256) void blurT_sp_kernel(const float* __restrict__ g, const float* __restrict__ kern,
                                                       const float* __restrict__ scale, int scale_stride,
                                                       const float* __restrict__ mul2, uint4* __restrict__ out, int C, int H,
                                                       int W, SPDims sp, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float lin[16 * BT_R * BT_C];     // 50688 B; reused as [16][512] in phase C
    __shared__ float kf[16];
    const int tid = threadIdx.x;
    int w = blockIdx.x;
    const int tx = w % tiles_x; w /= tiles_x;
    const int ty = w % tiles_y; w /= tiles_y;
    const int kc = w % sp.KC;
    const int b = w / sp.KC;
    const int i0 = ty * 4, j0 = tx * 32;
    const int Hg = 2 * H, Wg = 2 * W;                      // size of g
    if (tid < 16) kf[tid] = kern[tid];                     // the op flips the kernel it is given; pass flip(k) (symmetric anyway)
    // ---- A
    const int gy0 = 2 * i0 - 2, gx0 = 2 * j0 - 4;
    for (int e = tid; e < 16 * BT_R * (BT_C / 4); e += 256) {
        const int c4 = e % (BT_C / 4);
        const int r = (e / (BT_C / 4)) % BT_R;
        const int ch = e / ((BT_C / 4) * BT_R);
        const int c = kc * 16 + ch;
        const int gy = gy0 + r, gx = gx0 + 4 * c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C && gy >= 0 && gy < Hg && gx >= 0 && gx + 3 < Wg)
            v = *reinterpret_cast<const float4*>(g + (((long)b * C + c) * Hg + gy) * Wg + gx);
        *reinterpret_cast<float4*>(lin + (ch * BT_R + r) * BT_C + 4 * c4) = v;
    }
    __syncthreads();
    // ---- B: g2[Y,X] = sum_{a,b} kflip[a][b] * g[Y+a-2, X+b-2], kflip[a][b] = kern[3-a][3-b]
    const int ch = tid >> 4, tq = tid & 15;
    const int yrow = tq >> 1, xh = tq & 1;                 // g2 row 2*i0 + yrow, cols 2*j0 + 32*xh .. +31
    float o[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) o[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float* row = lin + (ch * BT_R + yrow + a) * BT_C + 32 * xh + 2;   // col index of X+0-2 relative to gx0: X - 2*j0 + 4 - 2
        float win[35];
#pragma unroll
        for (int j = 0; j < 35; ++j) win[j] = row[j];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float kv = kf[(3 - a) * 4 + (3 - bb)];
#pragma unroll
            for (int j = 0; j < 32; ++j) o[j] += kv * win[j + bb];
        }
    }
    const int c_glob = kc * 16 + ch;
    const float sc_ = (c_glob < C ? (scale ? scale[(long)b * scale_stride + c_glob] : 1.f) : 0.f) * (mul2 ? mul2[1] : 1.f);
    __syncthreads();                                       // everyone done reading the input tile
    // ---- exchange: lst[ch][pos], pos = ((ph*4 + il)*32 + jl), ph = (Y&1)*2 + (X&1)
    float* lst = lin;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int Xl = 32 * xh + j;                        // 0..63
        const int ph = (yrow & 1) * 2 + (Xl & 1);
        const int pos = (ph * 4 + (yrow >> 1)) * 32 + (Xl >> 1);
        lst[ch * 512 + pos] = o[j] * sc_;
    }
    __syncthreads();
    // ---- C: thread -> positions tid and tid+256
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int pos = tid + rep * 256;
        const int ph = pos >> 7, il = (pos >> 5) & 3, jl = pos & 31;
        const int i = i0 + il, j = j0 + jl;
        if (i > H || j > W) continue;                      // phase images are (H+1) x (W+1)
        half8 h0, h1, l0, l1;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            float v = lst[cc * 512 + pos];
            // entries outside the (2H+1)x(2W+1) support are exact zeros by construction except the row/col 2H / 2W parity
            const int Y = 2 * i + (ph >> 1), X = 2 * j + (ph & 1);
            if (Y > 2 * H || X > 2 * W) v = 0.f;
            const _Float16 hh = (_Float16)v;
            const _Float16 ll = (_Float16)(v - (float)hh);
            if (cc < 8) { h0[cc] = hh; l0[cc] = ll; } else { h1[cc - 8] = hh; l1[cc - 8] = ll; }
        }
        half8* rec = reinterpret_cast<half8*>(out + ((((long)b * sp.KC + kc) * 4 + ph) * sp.plane + ((long)i * sp.Wq + j) * 4));
        rec[0] = h0; rec[1] = h1; rec[2] = l0; rec[3] = l1;
    }
}

// F-form (B,C,H,W) fp32 -> S-form, value = x*scale[b,c]*mul2[1]; one thread per (b,kc,y,x) record
__global__ __launch_bounds__(256) void to_sform_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       int scale_stride, const float* __restrict__ shift, int shift_stride,
                                                       const float* __restrict__ mul2, uint4* __restrict__ out,
                                                       int B, SDims d, int in_pitch, unsigned* __restrict__ vmax) {
    const long total = (long)B * d.KC * d.H * d.W;
    const float gm = mul2 ? mul2[1] : 1.f;
    const long in_plane = (long)d.H * in_pitch;
    float vm = 0.f;
    int vb = -1;
    // uniform trip count per wave (the range-control flushes below use cross-lane operations)
    for (long base = blockIdx.x * 256L; base < total; base += (long)gridDim.x * 256) {
        const long e = base + threadIdx.x;
        const bool live = e < total;
        const int xx = (int)(e % d.W);
        const int yy = (int)((e / d.W) % d.H);
        const int kc = (int)((e / ((long)d.W * d.H)) % d.KC);
        const int b = live ? (int)(e / ((long)d.W * d.H * d.KC)) : vb;
        if (vmax && __any(b != vb)) {       // forward range control: max |v| per sample; a wave changes sample B-1 times
            record_vmax_mixed(vmax, vb, vm);
            vb = b; vm = 0.f;
        }
        if (!live) continue;
        half8 h0, h1, l0, l1;
        // the 16 channels of the record and their scales in one request (clamped channel; x as a valid address for absent scales):
        // `if (c < C) v = x[..] * scale[c]` is a branch and a wait per channel — sixteen round trips per record
        float xv[16], scv[16], shv[16];
        const float* scp = scale ? scale + (long)b * scale_stride : x;
        const float* shp = shift ? shift + (long)b * shift_stride : x;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int cc = min(kc * 16 + j, d.C - 1);
            xv[j] = x[((long)b * d.C + cc) * in_plane + (long)yy * in_pitch + xx];
            scv[j] = scp[cc];
            shv[j] = shp[cc];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = kc * 16 + j;
            float v = 0.f;
            if (c < d.C) v = (xv[j] * (scale ? scv[j] : 1.f) + (shift ? shv[j] : 0.f)) * gm;
            vm = fmaxf(vm, fabsf(v));
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)(v - (float)h);
            if (j < 8) { h0[j] = h; l0[j] = l; } else { h1[j - 8] = h; l1[j - 8] = l; }
        }
        const long u = sform_unit(d, b, kc, yy, xx, 0);
        half8* o = reinterpret_cast<half8*>(out + u);
        o[0] = h0; o[1] = h1; o[2] = l0; o[3] = l1;
    }
    if (vmax) record_vmax_mixed(vmax, vb, vm);
}

}  // namespace

extern "C" long oodgan_sform_bytes(int B, int C, int H, int W) {
    const SDims d = sform_dims(C, H, W);
    return (long)B * d.KC * d.plane * 16;
}

extern "C" int oodgan_to_sform(const float* x, const float* scale, int scale_stride, const float* shift, int shift_stride,
                               const float* mul2, void* out, int B, int C, int H, int W, int in_pitch, unsigned* vmax, void* stream) {
    OODGAN_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "to_sform: bad args");
    const SDims d = sform_dims(C, H, W);
    if (in_pitch == 0) in_pitch = W;
    const long total = (long)B * d.KC * H * W;
    hipLaunchKernelGGL(to_sform_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), x, scale, scale_stride,
                       shift, shift_stride, mul2, reinterpret_cast<uint4*>(out), B, d, in_pitch, vmax);
    return check_launch("to_sform");
}



namespace {
// S-form -> fp32 NCHW, divided by the scale that went into the S-form: the rarely taken way back for an activation that was saved only
// as its consumer's S-form (oodgan_conv_args.dotx_sform) when a step falls back to the two-pass backward.  thread = one 64-byte record
__global__ __launch_bounds__(256) void from_sform_kernel(const uint4* __restrict__ xs, const float* __restrict__ scale, int scale_stride,
                                                         float* __restrict__ y, int B, SDims d) {
    const long total = (long)B * d.KC * d.H * d.W;
    const long HW = (long)d.H * d.W;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % d.W), yy = (int)((e / d.W) % d.H);
        const int kc = (int)((e / HW) % d.KC), b = (int)(e / (HW * d.KC));
        const half8* rec = reinterpret_cast<const half8*>(xs + sform_unit(d, b, kc, yy, x, 0));
        const half8 h0 = rec[0], h1 = rec[1], l0 = rec[2], l1 = rec[3];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = kc * 16 + j;
            if (c < d.C) {
                const float v = (float)(j < 8 ? h0[j] : h1[j - 8]) + (float)(j < 8 ? l0[j] : l1[j - 8]);
                y[((long)b * d.C + c) * HW + (long)yy * d.W + x] = v / (scale ? scale[(long)b * scale_stride + c] : 1.f);
            }
        }
    }
}
}  // namespace

extern "C" int oodgan_from_sform(const void* xs, const float* scale, int scale_stride, float* y, int B, int C, int H, int W, void* stream) {
    OODGAN_REQUIRE(xs && y && B > 0 && C > 0 && H > 0 && W > 0, "from_sform: bad args");
    const SDims d = sform_dims(C, H, W);
    hipLaunchKernelGGL(from_sform_kernel, dim3(stream_grid((long)B * d.KC * H * W, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const uint4*>(xs), scale, scale_stride, y, B, d);
    return check_launch("from_sform");
}

extern "C" int oodgan_blurT_to_sform_phases(const float* g, const float* kernel, const float* scale, int scale_stride,
                                            const float* mul2, void* out, int B, int C, int H, int W, void* stream) {
    OODGAN_REQUIRE(g && kernel && out && B > 0 && C > 0 && H > 0 && W > 0, "blurT_to_sform_phases: bad args");
    OODGAN_REQUIRE((W % 2) == 0 && ((reinterpret_cast<uintptr_t>(g) & 15) == 0), "blurT_to_sform_phases: needs even W and an aligned input");
    const SPDims d = sp_dims(C, H, W);
    const int tiles_x = (W + 1 + 31) / 32, tiles_y = (H + 1 + 3) / 4;
    const long nb = (long)tiles_x * tiles_y * d.KC * B;
    OODGAN_REQUIRE(nb < (1L << 31), "blurT_to_sform_phases: grid too large");
    hipLaunchKernelGGL(blurT_sp_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), g, kernel, scale, scale_stride, mul2,
                       reinterpret_cast<uint4*>(out), C, H, W, d, tiles_x, tiles_y);
    return check_launch("blurT_to_sform_phases");
}

extern "C" long oodgan_sform_phases_bytes(int B, int C, int H, int W) {
    const SPDims d = sp_dims(C, H, W);
    return (long)B * d.KC * 4 * d.plane * 16;
}

static int to_sform_phases_impl(const float* x, const float* scale, int scale_stride, const float* mul2, void* out, int B,
                                int C, int H, int W, int in_pitch, int pad, void* stream) {
    OODGAN_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "to_sform_phases: bad args");
    const SPDims d = sp_dims(C, H, W);
    const int Hin = 2 * H + 1 - pad, Win = 2 * W + 1 - pad;
    if (in_pitch == 0) in_pitch = Win;
    const long total = (long)B * d.KC * 4 * (H + 1) * (W + 1);
    hipLaunchKernelGGL(to_sform_phases_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), x, scale,
                       scale_stride, mul2, reinterpret_cast<uint4*>(out), B, C, H, W, Hin, Win, in_pitch, d, pad);
    return check_launch("to_sform_phases");
}

extern "C" int oodgan_to_sform_phases(const float* x, const float* scale, int scale_stride, const float* mul2, void* out, int B,
                                      int C, int H, int W, int in_pitch, void* stream) {
    return to_sform_phases_impl(x, scale, scale_stride, mul2, out, B, C, H, W, in_pitch, 0, stream);
}

// x is the unpadded (B,C,2H,2W) input of nn.Conv2d(3, stride 2, padding 1): the zero row / column on the top / left come from the kernel
extern "C" int oodgan_to_sform_phases_padtl(const float* x, const float* scale, int scale_stride, const float* mul2, void* out, int B,
                                            int C, int H, int W, int in_pitch, void* stream) {
    return to_sform_phases_impl(x, scale, scale_stride, mul2, out, B, C, H, W, in_pitch, 1, stream);
}

namespace oodgan {

int launch_s1v2(const oodgan_conv_args& a, const void* wpk16, const float* unscale, hipStream_t st) {
    KArgs p;
    p.a = a;
    p.w_unscale = unscale;
    p.ablate = 0;
    p.Hn = a.Hin; p.Wn = a.Win; p.Hout = a.Hin; p.Wout = a.Win;
    OODGAN_REQUIRE(a.in_scale == nullptr && a.in_shift == nullptr, "conv3x3 S-form input: scales are applied by the producer");
    if (p.a.out_pitch == 0) p.a.out_pitch = p.Wout;
    p.in_plane = 0;
    p.out_plane = (long)p.Hout * p.a.out_pitch;
    p.tiles_y = (p.Hn + TR - 1) / TR;
    p.tiles_x = (p.Wn + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    // few work items and a long K loop (the 4x4 ... 32x32 layers): at most one workgroup per CU, every stage is a bare
    // DMA latency and the tile's MFMAs all sit on one CU -> 32-channel M tiles (twice the workgroups) and two K chunks
    // per stage
    const long items64 = (long)p.tiles_x * p.tiles_y * a.B * ((a.M + 63) / 64);
    const bool deep = items64 <= 256 && a.K >= 64;
    // < 128 input channels (the 512² layer): the single-group 64-channel instance needs 256 registers (two workgroups per
    // CU); 32-channel M tiles fit three and are 7 % faster although the x tile is fetched once per M block
    const bool mt2 = a.M > 32 && !deep && a.K >= 128;
    const int MB = mt2 ? 64 : 32;
    p.mblocks = (a.M + MB - 1) / MB;
    if (a.dotx) {
        OODGAN_REQUIRE(a.dot_part != nullptr, "conv3x3: dotx without dot_part");
        OODGAN_REQUIRE(a.dot_nparts == p.tiles_x * p.tiles_y, "conv3x3 f16s S1: dot_nparts %d != %d", a.dot_nparts,
                       p.tiles_x * p.tiles_y);
        OODGAN_REQUIRE((reinterpret_cast<uintptr_t>(a.dotx) & 15) == 0 || (a.Win % 4), "conv3x3: unaligned dotx");
    }
    OODGAN_REQUIRE(a.y || a.ys || a.dotx, "conv3x3: no output requested");
    SConv sc;
    sc.xs = reinterpret_cast<const uint4*>(a.x);
    sc.xd = sform_dims(a.K, a.Hin, a.Win);
    sc.ys = reinterpret_cast<uint4*>(a.ys);
    sc.yd = sform_dims(a.M, p.Hout, p.Wout);
    sc.ys_scale = a.ys_scale;
    sc.ys_scale_stride = a.ys_scale_stride;
    OODGAN_REQUIRE(!a.ys || (a.M % 16 == 0), "conv3x3: S-form output needs M %% 16 == 0");
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    const int items = (int)total;
    // NG = 2: two anti-phase pipelines per 512-thread workgroup (1 workgroup per CU); NG = 1: independent 256-thread
    // workgroups, 2-3 per CU — better when K is small and the epilogue dominates (it then overlaps other blocks)
    const int ng = a.K >= 128 ? 2 : 1;
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
#define OODGAN_LAUNCH(MT_, NG_, CPS_)                                                                                    \
    {                                                                                                                    \
        constexpr int stage_ = XBYTES + 36 * 32 * MT_ * 16;                                                              \
        constexpr int sm = NG_ * (stage_ * CPS_ > group_bytes<MT_>() ? stage_ * CPS_ : group_bytes<MT_>());              \
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1v2_kernel<MT_, NG_, CPS_>), \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, sm), true);            \
        (void)once;                                                                                                      \
        dim3 grid((unsigned)((total + NG_ - 1) / NG_)), block(256 * NG_);                                                \
        hipLaunchKernelGGL((conv_f16s_s1v2_kernel<MT_, NG_, CPS_>), grid, block, sm, st, p, w16, items, sc);             \
    }
    if (deep) {
        constexpr int sm = RING * (XBYTES + 36 * 32 * 16);
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1ring_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, sm), true);
        (void)once;
        hipLaunchKernelGGL((conv_f16s_s1ring_kernel<1>), dim3((unsigned)total), dim3(256), sm, st, p, w16, items, sc);
    }
    else if (mt2) { if (ng == 2) OODGAN_LAUNCH(2, 2, 1) else OODGAN_LAUNCH(2, 1, 1) }
    else { if (ng == 2) OODGAN_LAUNCH(1, 2, 1) else OODGAN_LAUNCH(1, 1, 1) }
#undef OODGAN_LAUNCH
    return check_launch("conv3x3_f16s_s1v2");
}

int launch_t2v2(const oodgan_conv_args& a, const void* wpk16, const float* unscale, hipStream_t st) {
    KArgs p;
    p.a = a;
    p.w_unscale = unscale;
    p.ablate = 0;
    p.Hn = a.Hin + 1; p.Wn = a.Win + 1; p.Hout = 2 * a.Hin + 1; p.Wout = 2 * a.Win + 1;
    OODGAN_REQUIRE(a.in_scale == nullptr && a.in_shift == nullptr, "conv3x3 S-form input: scales are applied by the producer");
    if (p.a.out_pitch == 0) p.a.out_pitch = p.Wout;
    OODGAN_REQUIRE((p.a.out_pitch & 1) == 0, "conv3x3 T2: out_pitch must be even (got %d)", p.a.out_pitch);
    OODGAN_REQUIRE(a.dotx == nullptr && a.noise == nullptr && a.bias == nullptr && a.act == OODGAN_ACT_NONE && a.ys == nullptr,
                   "conv3x3 T2: only out_scale is supported in the epilogue");
    p.in_plane = 0;
    p.out_plane = (long)p.Hout * p.a.out_pitch;
    p.tiles_y = (p.Hn + 3) / 4;
    p.tiles_x = (p.Wn + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    // few tiles (low-resolution layers): 32-channel M tiles put twice as many CUs to work on the same K loop
    // 32-channel M tiles: the 64-channel instance needs 276 registers (accumulators of 4 output phases x 2 M tiles), i.e.
    // one workgroup per CU; the 32-channel one fits three (140 registers, 32 KB of LDS) and is 30-35 % faster
    const int MB = 32;
    p.mblocks = (a.M + MB - 1) / MB;
    SConv sc;
    sc.xs = reinterpret_cast<const uint4*>(a.x);
    sc.xd = sform_dims(a.K, a.Hin, a.Win);
    sc.ys = nullptr; sc.yd = sc.xd; sc.ys_scale = nullptr; sc.ys_scale_stride = 0;
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    dim3 grid((unsigned)total), block(256);
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
    constexpr int sm = T2_XBYTES + 36 * 32 * 16;
    hipLaunchKernelGGL((conv_f16s_t2v2_kernel<1>), grid, block, sm, st, p, w16, sc);
    return check_launch("conv3x3_f16s_t2v2");
}

int launch_s2v2(const oodgan_conv_args& a, const void* wpk16, const float* unscale, hipStream_t st) {
    KArgs p;
    p.a = a;
    p.w_unscale = unscale;
    p.ablate = 0;
    p.Hn = (a.Hin - 1) / 2; p.Wn = (a.Win - 1) / 2; p.Hout = p.Hn; p.Wout = p.Wn;
    OODGAN_REQUIRE(a.in_scale == nullptr && a.in_shift == nullptr, "conv3x3 S-form input: scales are applied by the producer");
    OODGAN_REQUIRE(a.ys == nullptr, "conv3x3 S2: S-form output not supported");
    if (p.a.out_pitch == 0) p.a.out_pitch = p.Wout;
    p.in_plane = 0;
    p.out_plane = (long)p.Hout * p.a.out_pitch;
    p.tiles_y = (p.Hn + 7) / 8;
    p.tiles_x = (p.Wn + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    const bool mt2 = a.M > 32 && (long)p.tiles_x * p.tiles_y * a.B * ((a.M + 63) / 64) > 256;    // as in launch_t2v2
    const int MB = mt2 ? 64 : 32;
    p.mblocks = (a.M + MB - 1) / MB;
    if (a.dotx) {
        OODGAN_REQUIRE(a.dot_part != nullptr, "conv3x3: dotx without dot_part");
        OODGAN_REQUIRE(a.dot_nparts == p.tiles_x * p.tiles_y, "conv3x3 f16s S2: dot_nparts %d != %d", a.dot_nparts,
                       p.tiles_x * p.tiles_y);
    }
    SConv sc;
    sc.xs = reinterpret_cast<const uint4*>(a.x);
    sc.xd = sform_dims(a.K, p.Hn, p.Wn);
    sc.ys = nullptr; sc.yd = sc.xd; sc.ys_scale = nullptr; sc.ys_scale_stride = 0;
    const SPDims sp = sp_dims(a.K, p.Hn, p.Wn);
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    const int items = (int)total;
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
#define OODGAN_LAUNCH(MT_, NG_)                                                                                    \
    {                                                                                                              \
        constexpr int sm = NG_ * s2_group_bytes<MT_>();                                                            \
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2v2_kernel<MT_, NG_>),  \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, sm), true);      \
        (void)once;                                                                                                \
        dim3 grid((unsigned)((total + NG_ - 1) / NG_)), block(256 * NG_);                                          \
        hipLaunchKernelGGL((conv_f16s_s2v2_kernel<MT_, NG_>), grid, block, sm, st, p, w16, items, sc, sp);         \
    }
    // few tiles (low-resolution layers, already on 32-channel M tiles): one stage per channel chunk instead of two
    if (!mt2 && (long)p.tiles_x * p.tiles_y * a.B * ((a.M + 63) / 64) <= 256 && a.K >= 64) {
        constexpr int sm = 2 * S2_XBYTES + 9 * 4 * 32 * 16;
        static bool once1 = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2v2_kernel<1, 1, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, sm), true);
        (void)once1;
        hipLaunchKernelGGL((conv_f16s_s2v2_kernel<1, 1, true>), dim3((unsigned)total), dim3(256), sm, st, p, w16, items, sc, sp);
        return check_launch("conv3x3_f16s_s2v2");
    }
    if (mt2) OODGAN_LAUNCH(2, 2) else OODGAN_LAUNCH(1, 2)
#undef OODGAN_LAUNCH
    return check_launch("conv3x3_f16s_s2v2");
}

}  // namespace oodgan
