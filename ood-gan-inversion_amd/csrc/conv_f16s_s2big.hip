// Split-f16 3x3 STRIDE-2 conv on the phase-split S-form: the input gradient of the up-sampling ModulatedConv2d
// (adjoint of conv_transpose2d(stride 2), reference src/ops/StyleGAN/model.py:247-258) for the mid- and high-resolution
// layers, on the "all waves compute" structure of conv_f16s_big.hip.
//
//   dx[m,i,j] = sum_{k,ky,kx} g2[k, 2i+ky, 2j+kx] W[ky,kx]  =  stride-1 taps on the four parity images
//   G_{py,px}[i,j] = g2[2i+py, 2j+px]:  G_{py,px}[i+a, j+b] W[2a+py, 2b+px]   (a = 1 only for py = 0, b = 1 only for px = 0)
//
// One workgroup of 8 waves owns an 8 x 32 output tile and 64*MH output channels (MH = 2: four row pairs x two channel
// halves, every wave 2 rows x 2 M-tiles; MH = 1: eight rows, every wave 1 row x 2 M-tiles).  The K loop runs over stages
// (16-channel chunk, row parity py):
//   stage A (py = 0): x = 9 rows x {px 0,1} x 33 records (38 KB) + the 6 taps ky in {0,2} (4 KB * MH each)
//   stage B (py = 1): x = the same footprint of G_{1,*}              + the 3 taps ky = 1
// A and B stages alternate, so giving each its own LDS slot IS double buffering: the fetch of stage t+2 (LDS-DMA,
// global_load_lds) runs under the MFMAs of stage t+1, counted vmcnt waits, two barriers per stage.  The previous kernel
// (conv_f16s_s2v2_kernel<2,2>: two anti-phase 4-wave groups) had one group issuing MFMAs at a time and exposed one DMA
// latency per stage: 19 % MFMA-busy (profiles/, round 1).  With 128 channels per workgroup the x tile — four times the
// bytes of a stride-1 tile per output pixel — is fetched once per 128 instead of once per 64 output channels.
//   * x records keep the slot rotation (c>>2)&3 of conv_f16s_big.hip (conflict-free ds_read_b128, applied through the DMA
//     source address); weights arrive in their packed order [tap][hi|lo][k-half][channel][8].
//   * register epilogue as conv_f16s_big.hip: out-scale and 128-byte row segments, style-gradient dot via DPP + LDS.
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>
#include <cstdlib>
#include <type_traits>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

#ifdef OODGAN_CLOCK_STAMP
// Diagnostic build only (make STAMP=1, tools/s2big_probe.py): shader-cycle stamps of the phases of a FUSE workgroup — [workgroup][wave 0|7][start, before the K loop,
// after it, after the saved activations of M-tile 0 are in registers, after M-tile 0, after M-tile 1, end]
__device__ unsigned long long* g_s2big_stamp = nullptr;
__device__ long g_s2big_stamp_n = 0;
#define S2_STAMP(i) do { if (FUSE && (wave == 0 || wave == 7)) stv[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S2_STAMP(i)
#endif

namespace {

constexpr int SB_ROWS = 9, SB_C = 33;
constexpr int SB_ROWSLOTS = 2 * SB_C * 4;                  // 264 16-byte slots per tile row (both column parities)
constexpr int SB_XSLOTS = SB_ROWS * SB_ROWSLOTS;           // 2376
constexpr int SB_XPIECES = 40;                             // 5 one-KiB pieces per wave; the tail pieces are harmless duplicates
constexpr int SB_XBYTES = SB_XPIECES * 1024;

struct SPDims2 { int KC, Hq, Wq; long plane; };            // as sp_dims() of conv_f16s_v2.hip

__host__ __device__ inline SPDims2 sp_dims2(int C, int H, int W) {
    SPDims2 d;
    d.KC = (C + 15) / 16;
    d.Hq = (H + 7) / 8 * 8 + 2;
    d.Wq = (W + 31) / 32 * 32 + 2;
    d.plane = (long)d.Hq * d.Wq * 4;
    return d;
}

struct S2Big {
    oodgan_conv_args a;
    const uint4* xs;
    SPDims2 sp;
    const float* w_unscale;
    int Hout, Wout, tiles_x, tiles_y, mblocks, Mp;
    long out_plane;
    int ablate;      // profiling builds (-DOODGAN_DEBUG_ABLATE): 1 skip MFMAs, 2 skip the per-stage DMA
    oodgan_actbwd_fuse f;     // FUSE instances: the activation backward of the layer below (include/oodgan.h)
    SDims yd;                 // S-form of (M, Hout, Wout)
};

template <int MH>
struct S2Cfg {
    static constexpr int RW = MH == 2 ? 2 : 1;             // output rows per wave
    static constexpr int NRG = 8 / RW;                     // row groups
    static constexpr int MBW = 64 * MH;                    // channels per workgroup
    static constexpr int TAPB = 4 * MBW * 16;              // bytes per tap: [hi|lo][k-half][MBW][16 B]
    static constexpr int PPT = TAPB / 1024;                // DMA pieces per tap (8 / 4)
    static constexpr int WA = 6 * PPT / 8, WB = (3 * PPT + 7) / 8;      // weight pieces per wave and stage (6,3 / 3,2)
    static constexpr int NA = 5 + WA, NB = 5 + WB;         // loads per wave in flight for a stage
    static constexpr int OFF_XA = 0, OFF_WA = SB_XBYTES, OFF_XB = OFF_WA + 6 * TAPB, OFF_WB = OFF_XB + SB_XBYTES;
    static constexpr int SMEM = OFF_WB + 3 * TAPB;         // 155648 (MH = 2) / 118784 (MH = 1)
    // FUSE: [7][MBW] floats of channel constants, staged BEFORE the K loop: outside the stage buffers (159232 bytes for MH = 2)
    static constexpr int OFF_CST = SMEM;
    static constexpr int SMEM_FUSE = SMEM + 8 * MBW * 4;      // row 7: 1 / dotx_scale (dotx_sform)
};

constexpr int vmcnt_imm(int n) { return ((n >> 4) & 3) << 14 | 0x0F70 | (n & 15); }

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// G2 (oodgan_conv_args.x_hi_only, input-gradient instances): x_hi * (w_hi + w_lo) — the lo half of the gradient operand is neither read from LDS
// nor multiplied: two matrix instructions per tap instead of three (conv_f16s_big.hip, precision 'f16s-g2')
template <bool DOT, int MH, bool FUSE = false, bool G2 = false>
__global__ __launch_bounds__(512) void conv_f16s_s2big_kernel(const S2Big p, const uint4* __restrict__ wpk16) {
    using C = S2Cfg<MH>;
    constexpr int RW = C::RW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int rg = wave % C::NRG, mh = wave / C::NRG;      // row group, channel half

#ifdef OODGAN_CLOCK_STAMP
    unsigned long long stv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    S2_STAMP(0);
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int mblk = w % p.mblocks;
    w /= p.mblocks;
    const int ntile = p.tiles_x * p.tiles_y;
    const int tile = w % ntile, b = w / ntile;
    const int ty = tile / p.tiles_x, tx = tile % p.tiles_x;
    const int r0 = ty * 8, c0 = tx * 32, m0 = mblk * C::MBW;
    const int H = p.Hout, W = p.Wout, M = a.M;

    // ---- per-lane DMA source offsets (bytes) of this wave's 5 x pieces, relative to the (b, chunk, py) base
    // XH (G2 instances with oodgan_conv_args.x_hi_only == 2): x holds 32-byte hi-only records (oodgan_act_bwd_blurT_sform_phases_hi) — record r of a
    // phase plane at byte r*32 of the plane's first half.  The LDS image is unchanged (64-byte records whose lo slots stay unwritten: this instance never
    // reads them); the lanes of a DMA piece that would carry a lo slot are switched off: half the bytes from L2 / HBM, the same number of instructions
    const bool xh = G2 && a.x_hi_only == 2;
    unsigned offx[5];
    bool xok[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int P = (wave + 8 * i) * 64 + lane;
        if (P >= SB_XSLOTS) P = SB_XSLOTS - 1;
        const int row = P / SB_ROWSLOTS, rem = P % SB_ROWSLOTS;
        const int px = rem / (SB_C * 4), q = rem % (SB_C * 4);
        const int c = q >> 2, s = ((q & 3) - ((c >> 2) & 3)) & 3;
        if (xh) {
            // (the phase planes keep their stride: the 32-byte records fill the first half of each)
            offx[i] = (unsigned)(((long)px * p.sp.plane + ((long)(r0 + row) * p.sp.Wq + (c0 + c)) * 2 + (s & 1)) * 16);
            xok[i] = s < 2;
        } else {
            offx[i] = (unsigned)(((long)px * p.sp.plane + ((long)(r0 + row) * p.sp.Wq + (c0 + c)) * 4 + s) * 16);
            xok[i] = true;
        }
    }
    const long plane_bytes = p.sp.plane * 16;
    // grouped convolution (oodgan_conv_args.groups, plain epilogue only): the channel block selects its group's K input channels
    // of the G*K the S-form holds (p.sp.KC counts all of them)
    const int grp = a.groups > 1 ? m0 / (M / a.groups) : 0;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xs) + ((long)b * p.sp.KC + (long)grp * ((a.K + 15) / 16)) * 4 * plane_bytes;
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk16) + (long)m0 * 16;     // uniform; the lane adds lane16
    const unsigned lane16 = lane * 16;
    const long wchunk_bytes = (long)36 * p.Mp * 16;
    const long wrow_bytes = (long)p.Mp * 16;
    const int nchunk = (a.K + 15) / 16;
    const int nstage = 2 * nchunk;

    // stage st = (chunk t = st>>1, py = st&1); weights: py 0 -> taps ky in {0,2} (LDS slot (ky>>1)*3 + kx), py 1 -> ky = 1.
    // Piece j of this wave: j < 5 -> x piece, else weight piece j-5.  Pieces are issued one at a time between the MFMA groups
    // of the previous stage (a burst of 8-11 global_load_lds right after the barrier stalls every wave of the CU at once).
    auto dma_piece = [&](int st, int j) {
        const int t = st >> 1, py = st & 1;
        if (j < 5) {
            unsigned char* dx_ = smem + (py ? C::OFF_XB : C::OFF_XA);
            const unsigned char* xsrc = xb + ((long)t * 4 + py * 2) * plane_bytes;
            if (!G2 || xok[j])
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc + offx[j]),
                                                 (lds_void*)(dx_ + (wave + 8 * j) * 1024), 16, 0, 0);
            return;
        }
        const int i = j - 5;
        const int nw = py ? C::WB : C::WA, npieces = (py ? 3 : 6) * C::PPT;
        if (i >= nw) return;
        unsigned char* dw_ = smem + (py ? C::OFF_WB : C::OFF_WA);
        const unsigned char* wsrc = wb + (long)t * wchunk_bytes;
        int pw = wave + 8 * i;
        if (pw >= npieces) pw -= 8;                        // MH = 1, stage B: waves 4-7 repeat a piece (uniform load count)
        const int sl = pw / C::PPT, q = pw % C::PPT;
        const int tap = py ? 3 + sl : (sl / 3) * 6 + sl % 3;
        const int row = (q * 64) / C::MBW, j0 = (q * 64) % C::MBW;      // [hi|lo][k-half] row and channel offset of the piece
        const unsigned char* sb = wsrc + (long)(tap * 4 + row) * wrow_bytes + j0 * 16;      // scalar base + 32-bit lane offset
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + lane16), (lds_void*)(dw_ + pw * 1024), 16, 0, 0);
    };
    auto dma_stage = [&](int st) {
#pragma unroll
        for (int j = 0; j < C::NA; ++j) dma_piece(st, j);
    };

    f32x16 acc[2][RW];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < RW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // lane-constant fragment offsets: x column shift dxs in {0,1}; weights: lane -> (k-half, channel)
    unsigned lrd[2][2];
#pragma unroll
    for (int dxs = 0; dxs < 2; ++dxs)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
            const int c = dxs + l31;
            lrd[dxs][lo] = c * 64 + (((half + 2 * lo + ((c >> 2) & 3)) & 3) << 4);
        }
    const unsigned lwf = (half * C::MBW + mh * 64 + l31) * 16;

    // One stage = 3 (py = 1) or 6 (py = 0) taps on the SAME accumulators.  The fragments of tap i+1 are fetched from LDS while
    // the 12 MFMAs of tap i issue (two register sets, order pinned with sched_barrier): a wave's MFMA stream does not wait
    // for LDS, and both waves of a SIMD can keep the matrix pipe busy.
    struct Frag { half8 ah[2], al[2], bh[RW], bl[RW]; };
    auto load_tap = [&](Frag& f, const unsigned char* lx, const unsigned char* lw, auto sl_c) {
        constexpr int sl = decltype(sl_c)::value;
        constexpr int arow = sl / 3, kx = sl % 3;          // x row shift (ky = 2*arow in stage A, ky = 1 in stage B), column tap
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f.ah[mt] = *reinterpret_cast<const half8*>(lw + sl * C::TAPB + (0 * 2) * C::MBW * 16 + mt * 32 * 16);
            f.al[mt] = *reinterpret_cast<const half8*>(lw + sl * C::TAPB + (1 * 2) * C::MBW * 16 + mt * 32 * 16);
        }
#pragma unroll
        for (int nt = 0; nt < RW; ++nt) {
            const unsigned char* xr = lx + ((rg * RW + nt + arow) * 2 + (kx & 1)) * (SB_C * 64);
            f.bh[nt] = *reinterpret_cast<const half8*>(xr + lrd[kx >> 1][0]);
            if (!G2) f.bl[nt] = *reinterpret_cast<const half8*>(xr + lrd[kx >> 1][1]);
        }
    };
    auto mfma_tap = [&](const Frag& f) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < RW; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);
        if (!G2) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < RW; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mt], f.bl[nt], acc[mt][nt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < RW; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[mt], f.bh[nt], acc[mt][nt], 0, 0, 0);
    };
#define S2B_IC(n) std::integral_constant<int, n>{}
#define S2B_SB() __builtin_amdgcn_sched_barrier(0)

    // ---- FUSE, MH = 1: everything the fused epilogue reads from global memory is requested BEFORE the K loop — the channel
    // constants (-> LDS), the ToRGB gradient / noise of this wave's pixels and the saved activations of both M-tiles
    // (32 registers) — so that their latency runs under the K loop.  Measured on the 32 -> 64 channel layer @1024² -> 512²
    // (two 16-channel chunks: a K loop of 4 short stages), loads issued after the loop: K loop alone 427 us, epilogue alone
    // 891 us, together 1128 us; with the early loads 1010 us (in the W+ loop 1059 -> 923 us).  MH = 2 has no registers left
    // for this (it already spills; routing both forms through shared code cost it 16 more spilled registers and 8 % of
    // its time) and keeps its loads in the epilogue: the two forms are written out separately below.
    constexpr bool EARLY = FUSE && MH == 1;
    float eq0 = 0.f, eq1 = 0.f, eq2 = 0.f, enz = 0.f, edv[EARLY ? 2 : 1][16];
    bool eok = false;
    if constexpr (EARLY) {
        static_assert(!EARLY || RW == 1, "one row per wave");
        const oodgan_actbwd_fuse& f = p.f;
        const int px = c0 + l31, py_ = r0 + rg;
        const long HW = (long)H * W, pix = (long)py_ * W + px;
        eok = py_ < H && px < W;
        if (eok) {
            if (f.g_rgb) {
                const float* gr = f.g_rgb + (long)b * 3 * HW + pix;
                eq0 = gr[0]; eq1 = gr[HW]; eq2 = gr[2 * HW];
            }
            if (f.noise) enz = (f.noise_w ? f.noise_w[0] : 1.f) * f.noise[(long)(f.noise_batch > 1 ? b : 0) * HW + pix];
        }
        if (a.dotx_sform) {
            // the saved activation as the S-form its producer wrote: per 16-channel block four 8-byte pieces (hi / lo of channels
            // 4 half .. +3 and 8 + 4 half .. +3) — the same 32 registers, decoded in the epilogue
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const int kc = min((m0 + mh * 64 + mt * 32) / 16 + blk, p.yd.KC - 1);
                    const unsigned char* rec = reinterpret_cast<const unsigned char*>(a.dotx) +
                        sform_unit(p.yd, b, kc, min(py_, H - 1), min(px, W - 1), 0) * 16 + 8 * half;
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const float2 t = *reinterpret_cast<const float2*>(rec + 16 * sl);
                        edv[EARLY ? mt : 0][8 * blk + 2 * sl] = t.x;
                        edv[EARLY ? mt : 0][8 * blk + 2 * sl + 1] = t.y;
                    }
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float* dr = a.dotx + ((long)b * M + m0 + mh * 64 + mt * 32) * HW + pix;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
                    edv[EARLY ? mt : 0][r] = (eok && m0 + mh * 64 + mt * 32 + ch < M) ? dr[(long)ch * HW] : 0.f;
                }
            }
        }
    }

    // FUSE: the seven per-channel constants of the fused epilogue, requested here from clamped channels — all in flight together and
    // under the first stage's fetch — and parked in LDS.  (`mok ? ptr[m] : 0` where it is used is a branch, a load and a vmcnt(0) per
    // constant: seven serialised round trips per workgroup, 3-4 us of the 21 us a tile of the 1024² -> 512² layer takes.)
    float cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f};
    if constexpr (FUSE) {
        const oodgan_actbwd_fuse& f = p.f;
        if (tid < C::MBW) {
            const int m = min(m0 + tid, M - 1);
            cv[0] = a.out_scale ? a.out_scale[(long)b * a.out_scale_stride + m] : 1.f;
            if (f.g_rgb) {
                cv[1] = f.w_rgb[0 * M + m];
                cv[2] = f.w_rgb[1 * M + m];
                cv[3] = f.w_rgb[2 * M + m];
                cv[4] = f.s_rgb[(long)b * f.s_rgb_stride + m];
            }
            if (f.bias) cv[5] = f.bias[m];
            cv[6] = f.dscale[(long)b * f.dscale_stride + m];
            if (a.dotx_sform) {
                // a zero scale (style exactly 0): the S-form holds zeros and the activation cannot be recovered from it; decode as 0
                // instead of 0 * inf = NaN in the style-gradient sums
                const float sc = a.dotx_scale[(long)b * a.dotx_scale_stride + m];
                cv[7] = sc != 0.f ? 1.f / sc : 0.f;
            }
        }
    }
    // single barrier per stage: after it every wave has finished stage st-1 (its slot is free) and stage st has landed
    // (each wave waited for its own loads); the fetch of stage st+1 is issued first and runs under this stage's MFMAs
    dma_stage(0);
    if constexpr (FUSE) {
        if (tid < C::MBW) {
            const oodgan_actbwd_fuse& f = p.f;
            float* cst = reinterpret_cast<float*>(smem + C::OFF_CST);
            const bool mok = m0 + tid < M;
            cst[0 * C::MBW + tid] = mok ? cv[0] : 0.f;
            cst[1 * C::MBW + tid] = mok ? cv[1] * f.rgb_scale : 0.f;
            cst[2 * C::MBW + tid] = mok ? cv[2] * f.rgb_scale : 0.f;
            cst[3 * C::MBW + tid] = mok ? cv[3] * f.rgb_scale : 0.f;
            cst[4 * C::MBW + tid] = mok ? cv[4] : 0.f;
            cst[5 * C::MBW + tid] = mok ? cv[5] : 0.f;
            cst[6 * C::MBW + tid] = mok ? cv[6] * f.mul2[1] : 0.f;
            cst[7 * C::MBW + tid] = mok ? cv[7] : 0.f;
            __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): written before this wave arrives at the loop's first barrier
        }
    }
    S2_STAMP(1);
    for (int st = 0; st < nstage; ++st) {
        const int py = st & 1;
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        __builtin_amdgcn_s_barrier();
        const bool pf = st + 1 < nstage && !(p.ablate & 2);
        if (p.ablate & 1) { if (pf) dma_stage(st + 1); continue; }
        const unsigned char* lx = smem + (py ? C::OFF_XB : C::OFF_XA);
        const unsigned char* lw = smem + (py ? C::OFF_WB : C::OFF_WA) + lwf;
        Frag f0, f1;
        // MH = 2 runs at the 256-register limit: interleaving the pieces there costs spills (measured 261 -> 355 us), so it
        // keeps the burst right after the barrier; MH = 1 interleaves (139 -> 125 us on the 64² -> 32² layer)
        if (MH == 2 && pf) dma_stage(st + 1);
#define S2B_DMA(j) if (MH == 1 && pf) dma_piece(st + 1, (j));
        load_tap(f0, lx, lw, S2B_IC(0));
        S2B_SB();
        if (py == 0) {      // 6 taps; prefetch the 5 + WB pieces of the following py = 1 stage
            load_tap(f1, lx, lw, S2B_IC(1)); S2B_DMA(0) S2B_DMA(1) S2B_SB(); mfma_tap(f0); S2B_SB();
            load_tap(f0, lx, lw, S2B_IC(2)); S2B_DMA(2) S2B_DMA(3) S2B_SB(); mfma_tap(f1); S2B_SB();
            load_tap(f1, lx, lw, S2B_IC(3)); S2B_DMA(4) S2B_SB(); mfma_tap(f0); S2B_SB();
            load_tap(f0, lx, lw, S2B_IC(4)); S2B_DMA(5) S2B_SB(); mfma_tap(f1); S2B_SB();
            load_tap(f1, lx, lw, S2B_IC(5)); S2B_DMA(6) S2B_SB(); mfma_tap(f0); S2B_SB();
            S2B_DMA(7) S2B_SB();
            mfma_tap(f1);
        } else {            // 3 taps; prefetch the 5 + WA pieces of the following py = 0 stage
            load_tap(f1, lx, lw, S2B_IC(1)); S2B_DMA(0) S2B_DMA(1) S2B_DMA(2) S2B_DMA(3) S2B_SB(); mfma_tap(f0); S2B_SB();
            load_tap(f0, lx, lw, S2B_IC(2)); S2B_DMA(4) S2B_DMA(5) S2B_DMA(6) S2B_DMA(7) S2B_SB(); mfma_tap(f1); S2B_SB();
            S2B_DMA(8) S2B_DMA(9) S2B_DMA(10) S2B_SB();
            mfma_tap(f0);
        }
#undef S2B_DMA
    }
    __builtin_amdgcn_s_barrier();            // LDS is reused by the dot reduction below
    S2_STAMP(2);
#undef S2B_IC
#undef S2B_SB

    if constexpr (FUSE) {
        // ---- fused epilogue: style-gradient dot + activation backward of the layer below, written as its S-form gradient.
        // Per-channel constants of the workgroup's channels go through LDS (a lane needs 16 channels x 7 values per M-tile).
        static_assert(DOT, "the fused epilogue includes the dot");
        const oodgan_actbwd_fuse& f = p.f;
        float* red = reinterpret_cast<float*>(smem);                     // [row group][3][MBW]
        float* cst = reinterpret_cast<float*>(smem + C::OFF_CST);        // [7][MBW]: a*us->g_feat scale, w0, w1, w2, s_rgb, bias, d*scale
        const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
        const float rscale = f.mul2[1];
        const int px = c0 + l31;
        const long HW = (long)H * W;
        const float nw = f.noise ? (f.noise_w ? f.noise_w[0] : 1.f) : 0.f;
        // pixel-level inputs of this wave's rows: ToRGB gradient (3 colours) and the noise map
        float q0[RW], q1[RW], q2[RW], nzv[RW];
        bool okr[RW];
#pragma unroll
        for (int nt = 0; nt < RW; ++nt) {
            const int py_ = r0 + rg * RW + nt;
            okr[nt] = py_ < H && px < W;
            q0[nt] = q1[nt] = q2[nt] = nzv[nt] = 0.f;
            if constexpr (EARLY) {
                okr[nt] = eok; q0[nt] = eq0; q1[nt] = eq1; q2[nt] = eq2; nzv[nt] = enz;
            } else if (okr[nt]) {
                const long pix = (long)py_ * W + px;
                if (f.g_rgb) {
                    const float* gr = f.g_rgb + (long)b * 3 * HW + pix;
                    q0[nt] = gr[0]; q1[nt] = gr[HW]; q2[nt] = gr[2 * HW];
                }
                if (f.noise) nzv[nt] = nw * f.noise[(long)(f.noise_batch > 1 ? b : 0) * HW + pix];
            }
        }
        float vmaxv = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (MH == 2) __builtin_amdgcn_sched_barrier(0);                // the second M-tile's 32 loads stay below the first one's work
            const int chb = mh * 64 + mt * 32;                             // first channel of the M-tile inside the workgroup
            // saved activations of the tile: all loads of the M-tile are issued before the first use
            // Unconditional loads from a clamped pixel (M % 32 == 0: every channel of the tile exists; a conditional load is a
            // branch per value), out-of-image pixels masked by a select.  Address = uniform base + a uniform channel offset per
            // value + ONE 32-bit lane offset per row (sixteen 64-bit lane addresses per row would cost 32 registers).
            float dv[RW][16];
            if (a.dotx_sform) {
                // S-form records (see the early loads above): raw[8 blk + 2 sl + {0,1}] = slot sl of block blk — hi slots 0,1, lo slots 2,3,
                // each holding this lane's four channels as f16; value = (hi + lo) / scale
#pragma unroll
                for (int nt = 0; nt < RW; ++nt) {
                    float raw[16];
                    if constexpr (EARLY) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) raw[i] = edv[EARLY ? mt : 0][i];
                    } else {
                        const int py_ = r0 + rg * RW + nt;
#pragma unroll
                        for (int blk = 0; blk < 2; ++blk) {
                            const int kc = min((m0 + chb) / 16 + blk, p.yd.KC - 1);
                            const unsigned char* rec = reinterpret_cast<const unsigned char*>(a.dotx) +
                                sform_unit(p.yd, b, kc, min(py_, H - 1), min(px, W - 1), 0) * 16 + 8 * half;
#pragma unroll
                            for (int sl = 0; sl < 4; ++sl) {
                                const float2 t = *reinterpret_cast<const float2*>(rec + 16 * sl);
                                raw[8 * blk + 2 * sl] = t.x;
                                raw[8 * blk + 2 * sl + 1] = t.y;
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // register r: block r >> 3, slot (r >> 2) & 1, element r & 3 of the 8-byte piece
                        const int blk = r >> 3, sl = (r >> 2) & 1, el = r & 3;
                        const unsigned wh = __float_as_uint(raw[8 * blk + 2 * sl + (el >> 1)]), wl = __float_as_uint(raw[8 * blk + 2 * (2 + sl) + (el >> 1)]);
                        const unsigned short bh = (unsigned short)((el & 1) ? (wh >> 16) : (wh & 0xFFFFu)), bl = (unsigned short)((el & 1) ? (wl >> 16) : (wl & 0xFFFFu));
                        const float v = (float)__builtin_bit_cast(_Float16, bh) + (float)__builtin_bit_cast(_Float16, bl);
                        const float inv = cst[7 * C::MBW + chb + (r & 3) + 8 * (r >> 2) + 4 * half];
                        dv[nt][r] = okr[nt] ? v * inv : 0.f;
                    }
                }
            } else {
            const unsigned char* ub = reinterpret_cast<const unsigned char*>(a.dotx + ((long)b * M + m0 + chb) * HW);
#pragma unroll
            for (int nt = 0; nt < RW; ++nt) {
                const int py_ = r0 + rg * RW + nt;
                const unsigned voff = (unsigned)((((long)min(py_, H - 1) * W + min(px, W - 1)) + (long)(4 * half) * HW) * 4);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (EARLY) dv[nt][r] = edv[EARLY ? mt : 0][r];
                    else dv[nt][r] = *reinterpret_cast<const float*>(ub + (long)((r & 3) + 8 * (r >> 2)) * HW * 4 + voff);
                }
                if constexpr (!EARLY) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dv[nt][r] = okr[nt] ? dv[nt][r] : 0.f;
                }
            }
            }
#ifdef OODGAN_CLOCK_STAMP
            asm volatile("" : "+v"(dv[0][0]), "+v"(dv[0][15]));
            if (mt == 0) S2_STAMP(3);
#endif
            unsigned hi[RW][8], lo[RW][8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                float vv[RW][2];
                // registers 2 r2 and 2 r2 + 1 are two consecutive channels: their seven constants come as seven 8-byte LDS reads
                const int ch0 = chb + ((2 * r2) & 3) + 8 * ((2 * r2) >> 2) + 4 * half;
                float2 k7[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) k7[i] = *reinterpret_cast<const float2*>(cst + i * C::MBW + ch0);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * r2 + e;
                    const int ch = ch0 + e;
                    const float c_g = e ? k7[0].y : k7[0].x, w0 = e ? k7[1].y : k7[1].x, w1 = e ? k7[2].y : k7[2].x, w2 = e ? k7[3].y : k7[3].x;
                    const float srg = e ? k7[4].y : k7[4].x, bv = e ? k7[5].y : k7[5].x, ds = e ? k7[6].y : k7[6].x;
                    float ad = 0.f, ar = 0.f, at = 0.f;
#pragma unroll
                    for (int nt = 0; nt < RW; ++nt) {
                        const float a0 = acc[mt][nt][r] * us;              // raw conv sum = d(loss)/d(x*s) of the up-conv
                        const float o = dv[nt][r];
                        ad += a0 * o;
                        const float t = w0 * q0[nt] + w1 * q1[nt] + w2 * q2[nt];
                        const float g = a0 * c_g + srg * t;
                        const float gp = okr[nt] ? g * (o > 0.f ? kSqrt2 : 0.2f * kSqrt2) : 0.f;
                        const float ycv = (o > 0.f ? o * kInvPos : o * kInvNeg) - nzv[nt] - bv;
                        ar += gp * ycv;
                        at += o * t;
                        const float v = gp * ds;
                        vmaxv = fmaxf(vmaxv, fabsf(v));
                        vv[nt][e] = v;
                    }
                    // the three per-channel sums of this register: lanes -> row group -> LDS, right away (held in sd / sr / st arrays
                    // until the end of the M-tile they were 48 live registers: the MH = 2 instance spilled 7 of them, round 3)
                    const float v0 = half_sum_dpp(ad), v1 = half_sum_dpp(ar), v2 = half_sum_dpp(at);
                    if (l31 == kHalfSumLane) {
                        red[(rg * 3 + 0) * C::MBW + ch] = v0;
                        red[(rg * 3 + 1) * C::MBW + ch] = v1;
                        red[(rg * 3 + 2) * C::MBW + ch] = v2;
                    }
                }
                if (f.ys_hi_only) {       // the consumer never reads a lo half (x_hi_only = 2): no second conversion, no lo store
#pragma unroll
                    for (int nt = 0; nt < RW; ++nt) {
                        half2v h2;
                        h2[0] = (_Float16)vv[nt][0];
                        h2[1] = (_Float16)vv[nt][1];
                        hi[nt][r2] = __builtin_bit_cast(unsigned, h2);
                        lo[nt][r2] = 0u;
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < RW; ++nt) split_pair(vv[nt][0], vv[nt][1], hi[nt][r2], lo[nt][r2]);
                }
            }
            // 64-byte records: lanes 0-31 hold channels {0-3, 8-11} of a 16-channel block, lanes 32-63 {4-7, 12-15};
            // one v_permlane32_swap pair completes the 16-byte slots (half 0 -> slots 0 / 2, half 1 -> slots 1 / 3)
#pragma unroll
            for (int nt = 0; nt < RW; ++nt) {
                const int py_ = r0 + rg * RW + nt;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    auto h0 = __builtin_amdgcn_permlane32_swap(hi[nt][cb * 4 + 0], hi[nt][cb * 4 + 2], false, false);
                    auto h1 = __builtin_amdgcn_permlane32_swap(hi[nt][cb * 4 + 1], hi[nt][cb * 4 + 3], false, false);
                    const int kc = (m0 + chb) / 16 + cb;
                    if (f.ys_hi_only) {
                        if (okr[nt] && kc < p.yd.KC) {      // record at half the in-plane offset: 32 lanes x 2 halves = 1 KB contiguous per wave and row
                            uint4* rec = reinterpret_cast<uint4*>(f.ys) + (((long)b * p.yd.KC + kc) * p.yd.plane + ((long)(py_ + 1) * p.yd.Wp + (px + 1)) * 2);
                            rec[half] = make_uint4(h0[0], h1[0], h0[1], h1[1]);
                        }
                        continue;
                    }
                    auto l0 = __builtin_amdgcn_permlane32_swap(lo[nt][cb * 4 + 0], lo[nt][cb * 4 + 2], false, false);
                    auto l1 = __builtin_amdgcn_permlane32_swap(lo[nt][cb * 4 + 1], lo[nt][cb * 4 + 3], false, false);
                    if (okr[nt] && kc < p.yd.KC) {
                        uint4* rec = reinterpret_cast<uint4*>(f.ys) + sform_unit(p.yd, b, kc, py_, px, 0);
                        rec[half] = make_uint4(h0[0], h1[0], h0[1], h1[1]);
                        rec[2 + half] = make_uint4(l0[0], l1[0], l0[1], l1[1]);
                    }
                }
            }
#ifdef OODGAN_CLOCK_STAMP
            __builtin_amdgcn_sched_barrier(0);
            if (mt == 0) S2_STAMP(4); else S2_STAMP(5);
#endif
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmaxv = fmaxf(vmaxv, __shfl_xor(vmaxv, o, 64));
        if (lane == 0) f.part_max[(((long)b * ntile + tile) * p.mblocks + mblk) * 8 + wave] = vmaxv / rscale;   // max |g_pre*d|
        __syncthreads();
        if (tid < C::MBW && m0 + tid < M) {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int g = 0; g < C::NRG; ++g) {
                v0 += red[(g * 3 + 0) * C::MBW + tid];
                v1 += red[(g * 3 + 1) * C::MBW + tid];
                v2 += red[(g * 3 + 2) * C::MBW + tid];
            }
            const long o = ((long)b * M + m0 + tid) * a.dot_nparts + tile;
            a.dot_part[o] = v0;
            f.part_r[o] = v1;
            if (f.part_t) f.part_t[o] = v2;
        }
#ifdef OODGAN_CLOCK_STAMP
        S2_STAMP(6);
        if (lane == 0 && (wave == 0 || wave == 7) && g_s2big_stamp && (long)blockIdx.x < g_s2big_stamp_n) {
            unsigned long long* q = g_s2big_stamp + ((long)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
#pragma unroll
            for (int i = 0; i < 7; ++i) q[i] = stv[i];
            q[7] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        return;
    }
    // ---- epilogue from the accumulators (conv_f16s_big.hip's, NT = RW rows per wave)
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
    const int px = c0 + l31;
    float dsum[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dsum[mt][r] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float osc[16], bia[16], slp[16];
        unsigned moff[16], doff[16];
        // the constants of the M-tile from clamped channels, all in flight together (the packed weights as a valid address for an absent
        // tensor), masked by selects: `mok ? ptr[m] : 0` per register was a branch, a load and a vmcnt(0) each
        const float* dmy = reinterpret_cast<const float*>(wpk16);
        const float* oscp = a.out_scale ? a.out_scale + (long)b * a.out_scale_stride : dmy;
        const float* biap = (!DOT && a.bias) ? a.bias : dmy;
        const float* slpp = (!DOT && a.act == OODGAN_ACT_PRELU) ? a.slope : dmy;
        float c_o[16], c_b[16], c_s[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mc = min(m0 + mh * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, M - 1);
            c_o[r] = oscp[a.out_scale ? mc : 0];
            c_b[r] = biap[(!DOT && a.bias) ? mc : 0];
            c_s[r] = slpp[(!DOT && a.act == OODGAN_ACT_PRELU) ? mc : 0];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mh * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool mok = m < M;
            osc[r] = mok ? (a.out_scale ? c_o[r] : 1.f) * us : 0.f;
            bia[r] = (!DOT && a.bias && mok) ? c_b[r] : 0.f;
            slp[r] = (!DOT && a.act == OODGAN_ACT_PRELU && mok) ? c_s[r] : 1.f;
            moff[r] = mok ? (unsigned)((long)m * p.out_plane * 4) : 0xFFFFFFFFu;
            doff[r] = mok ? (unsigned)((long)m * H * W * 4) : 0u;
        }
        const bool mfull = m0 + mh * 64 + mt * 32 + 32 <= M;
#pragma unroll
        for (int nt = 0; nt < RW; ++nt) {
            const int py_ = r0 + rg * RW + nt;
            const bool ok = py_ < H && px < W;
            unsigned char* yr = reinterpret_cast<unsigned char*>(a.y) + ((long)b * M * p.out_plane + (long)py_ * a.out_pitch + px) * 4;
            const unsigned char* dr = DOT ? reinterpret_cast<const unsigned char*>(a.dotx) + ((long)b * M * H * W + (long)py_ * W + px) * 4 : nullptr;
            float dv[16];
            if (DOT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dv[r] = 0.f;
                if (ok) {
                    if (mfull) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dv[r] = *reinterpret_cast<const float*>(dr + doff[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (moff[r] != 0xFFFFFFFFu) dv[r] = *reinterpret_cast<const float*>(dr + doff[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dsum[mt][r] += (acc[mt][nt][r] * us) * dv[r];
            }
            if (ok && a.y) {
                float o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[r] = acc[mt][nt][r] * osc[r];
                    if (!DOT) {          // forward use (the stride-2 convs of the e4e encoder): bias + PReLU / leaky ReLU
                        o[r] += bia[r];
                        if (a.act == OODGAN_ACT_LRELU) o[r] = (o[r] > 0.f ? o[r] : 0.2f * o[r]) * kSqrt2;
                        else if (a.act == OODGAN_ACT_PRELU) o[r] = o[r] > 0.f ? o[r] : slp[r] * o[r];
                    }
                }
                if (mfull) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(yr + moff[r]) = o[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (moff[r] != 0xFFFFFFFFu) *reinterpret_cast<float*>(yr + moff[r]) = o[r];
                }
            }
        }
    }
    if (DOT) {
        // per-wave sums over its rows -> LDS [row group][channel of the workgroup] -> one partial per (b, m, tile)
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = half_sum_dpp(dsum[mt][r]);
                if (l31 == kHalfSumLane) red[rg * C::MBW + mh * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = v;
            }
        __syncthreads();
        if (tid < C::MBW && m0 + tid < M) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < C::NRG; ++g) v += red[g * C::MBW + tid];
            a.dot_part[((long)b * M + m0 + tid) * a.dot_nparts + tile] = v;
        }
    }
}

}  // namespace

namespace oodgan {

bool s2_big_eligible(const oodgan_conv_args& a) {
    if (!(a.mode == OODGAN_CONV_S2 && a.x_sform && a.ys == nullptr && a.in_scale == nullptr && a.in_shift == nullptr &&
          a.noise == nullptr && (a.y != nullptr || a.dotx != nullptr) && a.M >= 64))
        return false;
    // bias / activation: only in the plain (no dot, no fused backward) epilogue — the forward use by the encoder
    if ((a.bias != nullptr || a.act != OODGAN_ACT_NONE) && (a.dotx != nullptr || a.fuse != nullptr || a.y == nullptr)) return false;
    if (a.groups > 1 && (a.dotx != nullptr || a.fuse != nullptr || a.K % 16 != 0 || a.M % a.groups != 0 || (a.M / a.groups) % 128 != 0)) return false;
    if (a.fuse) return true;                // the fused epilogue exists only here (the caller checked s2_fuse_supported)
    // enough 8x32 tiles x 64-channel blocks to fill the chip; the low-resolution layers keep their latency-oriented instance
    const int Hn = (a.Hin - 1) / 2, Wn = (a.Win - 1) / 2;
    const long items = (long)((Hn + 7) / 8) * ((Wn + 31) / 32) * a.B * ((a.M + 63) / 64);
    // 128 work items = half the CUs: what a sub-batch of 2-3 images (three concurrent streams) brings to the 64² / 32² layers.  Whole loop,
    // 3 streams, same box: threshold 256 -> 5.67 img/s, 128 -> 5.79 (one stream, batch 8: the 32² input gradient 169 -> 150 us)
    return items >= tunable(OODGAN_TUN_S2_BIG_MIN_ITEMS);      // default 128; tests lower it to reach this kernel with small tensors     // 512 -> 512 @64² -> 32² (256 items): 229 -> 139 us against the merged-parity tile kernel
}

}  // namespace oodgan

// 1 when oodgan_conv3x3_f16s (mode S2, S-form input) of this shape runs the kernel that has the fused activation backward
extern "C" int oodgan_conv3x3_s2_fuse_supported(int B, int K, int M, int Hin, int Win) {
    oodgan_conv_args a = {};
    a.mode = OODGAN_CONV_S2; a.x_sform = 1; a.B = B; a.K = K; a.M = M; a.Hin = Hin; a.Win = Win;
    a.y = reinterpret_cast<float*>(1);
    return (M % 32) == 0 && oodgan::s2_big_eligible(a) ? 1 : 0;
}

extern "C" int oodgan_conv3x3_s2_grouped_supported(int B, int K, int M, int groups, int Hin, int Win) {
    oodgan_conv_args a = {};
    a.mode = OODGAN_CONV_S2; a.x_sform = 1; a.B = B; a.K = K; a.M = M; a.Hin = Hin; a.Win = Win; a.groups = groups;
    a.y = reinterpret_cast<float*>(1);
    return groups > 1 && oodgan::s2_big_eligible(a) ? 1 : 0;
}

#ifdef OODGAN_CLOCK_STAMP
extern "C" int oodgan_debug_set_s2big_stamp_buffer(void* buf, long n) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(buf);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_s2big_stamp), &q, sizeof(q)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_s2big_stamp_n), &n, sizeof(n)) != hipSuccess) {
        oodgan::set_error("debug_set_s2big_stamp_buffer: hipMemcpyToSymbol failed");
        return OODGAN_E_LAUNCH;
    }
    return 0;
}
#endif

namespace oodgan {

int launch_s2_big(const oodgan_conv_args& a_in, const void* wpk16, const float* unscale, hipStream_t st) {
    S2Big p;
    p.a = a_in;
    oodgan_conv_args& a = p.a;
    p.Hout = (a.Hin - 1) / 2;
    p.Wout = (a.Win - 1) / 2;
    if (a.out_pitch == 0) a.out_pitch = p.Wout;
    p.out_plane = (long)p.Hout * a.out_pitch;
    p.xs = reinterpret_cast<const uint4*>(a.x);
    p.sp = sp_dims2(a.K * (a.groups > 1 ? a.groups : 1), p.Hout, p.Wout);
    p.w_unscale = unscale;
    p.tiles_y = (p.Hout + 7) / 8;
    p.tiles_x = (p.Wout + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    // 128 channels per workgroup when the channel count allows it and the grid still covers the chip
    const long items128 = (long)p.tiles_x * p.tiles_y * a.B * (a.M / 128);
    const bool mh2 = (a.M % 128) == 0 && items128 >= 256;
    p.mblocks = mh2 ? a.M / 128 : (a.M + 63) / 64;
    if (a.dotx) {
        OODGAN_REQUIRE(a.dot_part != nullptr, "conv3x3: dotx without dot_part");
        OODGAN_REQUIRE(a.dot_nparts == p.tiles_x * p.tiles_y, "conv3x3 f16s S2: dot_nparts %d != %d", a.dot_nparts, p.tiles_x * p.tiles_y);
    }
    OODGAN_REQUIRE((long)a.M * p.out_plane * 4 < (1L << 32), "conv3x3 S2 big: plane too large");
    const bool fuse = a.fuse != nullptr;
    p.yd = sform_dims(a.M, p.Hout, p.Wout);
    if (fuse) {
        p.f = *a.fuse;
        OODGAN_REQUIRE(a.dotx && a.dot_part && p.f.ys && p.f.part_r && p.f.part_max && p.f.dscale && p.f.mul2 && (a.M % 32) == 0,
                       "conv3x3 S2 fused activation backward: needs dotx, ys, part_r, part_max, dscale, mul2 and M %% 32 == 0");
        OODGAN_REQUIRE(!p.f.g_rgb || (p.f.w_rgb && p.f.s_rgb && p.f.part_t), "conv3x3 S2 fused: rgb branch needs w_rgb, s_rgb and part_t");
        OODGAN_REQUIRE(p.f.noise == nullptr || p.f.noise_batch == 1 || p.f.noise_batch == a.B, "conv3x3 S2 fused: noise_batch");
        OODGAN_REQUIRE(p.f.nmax >= (long)a.B * p.tiles_x * p.tiles_y * p.mblocks * 8, "conv3x3 S2 fused: part_max too small");
    }
    OODGAN_REQUIRE(p.sp.plane * 32 < (1L << 32), "conv3x3 S2 big: input plane too large");
#ifdef OODGAN_DEBUG_ABLATE      // profiling builds only: the ablation bits make the kernel skip work (wrong results)
    static const int abl = getenv("OODGAN_S2BIG_ABLATE") ? atoi(getenv("OODGAN_S2BIG_ABLATE")) : 0;
    p.ablate = abl;
#else
    p.ablate = 0;
#endif
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<2>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<2>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<1>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<1>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<true, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<2>::SMEM_FUSE),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<true, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<1>::SMEM_FUSE),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<true, 2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<2>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<true, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<1>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<true, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<2>::SMEM_FUSE),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s2big_kernel<true, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Cfg<1>::SMEM_FUSE), true);
    (void)once;
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
    const bool g2 = a.x_hi_only != 0 && a.dotx != nullptr;
    OODGAN_REQUIRE(a.x_hi_only != 2 || (g2 && a.groups <= 1), "conv3x3 S2 big: hi-only input records (x_hi_only = 2) need dotx (the two-instruction instances) and no groups");
    if (g2) {
        count_dispatch(OODGAN_DC_S2BIG_G2);
        if (a.x_hi_only == 2) count_dispatch(OODGAN_DC_S2BIG_XH);
        if (fuse) {
            if (mh2) hipLaunchKernelGGL((conv_f16s_s2big_kernel<true, 2, true, true>), dim3((unsigned)total), dim3(512), S2Cfg<2>::SMEM_FUSE, st, p, w16);
            else hipLaunchKernelGGL((conv_f16s_s2big_kernel<true, 1, true, true>), dim3((unsigned)total), dim3(512), S2Cfg<1>::SMEM_FUSE, st, p, w16);
        } else if (mh2) hipLaunchKernelGGL((conv_f16s_s2big_kernel<true, 2, false, true>), dim3((unsigned)total), dim3(512), S2Cfg<2>::SMEM, st, p, w16);
        else hipLaunchKernelGGL((conv_f16s_s2big_kernel<true, 1, false, true>), dim3((unsigned)total), dim3(512), S2Cfg<1>::SMEM, st, p, w16);
    } else if (fuse) {
        if (mh2) hipLaunchKernelGGL((conv_f16s_s2big_kernel<true, 2, true>), dim3((unsigned)total), dim3(512), S2Cfg<2>::SMEM_FUSE, st, p, w16);
        else hipLaunchKernelGGL((conv_f16s_s2big_kernel<true, 1, true>), dim3((unsigned)total), dim3(512), S2Cfg<1>::SMEM_FUSE, st, p, w16);
    } else if (mh2) {
        if (a.dotx) hipLaunchKernelGGL((conv_f16s_s2big_kernel<true, 2>), dim3((unsigned)total), dim3(512), S2Cfg<2>::SMEM, st, p, w16);
        else hipLaunchKernelGGL((conv_f16s_s2big_kernel<false, 2>), dim3((unsigned)total), dim3(512), S2Cfg<2>::SMEM, st, p, w16);
    } else {
        if (a.dotx) hipLaunchKernelGGL((conv_f16s_s2big_kernel<true, 1>), dim3((unsigned)total), dim3(512), S2Cfg<1>::SMEM, st, p, w16);
        else hipLaunchKernelGGL((conv_f16s_s2big_kernel<false, 1>), dim3((unsigned)total), dim3(512), S2Cfg<1>::SMEM, st, p, w16);
    }
    return check_launch("conv3x3_f16s_s2big");
}

}  // namespace oodgan
