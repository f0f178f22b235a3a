// Split-f16 TRANSPOSED 3x3 stride-2 conv on an S-form input: the up-sampling ModulatedConv2d before its blur
// (conv_transpose2d(stride 2, padding 0), reference src/ops/StyleGAN/model.py:247-258) for the mid- and high-resolution
// layers, on the "all waves compute" structure of conv_f16s_big.hip.
//
//   z[m, 2i'+py, 2j'+px] = sum_{k, ky = py (mod 2), kx = px (mod 2)} W[m,k,ky,kx] * x[k, i' - (ky>>1), j' - (kx>>1)]
// over the (H+1) x (W+1) grid of positions (i', j'): 9 tap products per position feed 4 output phases (4 / 2 / 2 / 1 taps).
//
// One workgroup of 8 waves owns 8 position rows x 32 position columns and 64 output channels; every wave one row, two
// M-tiles and all four phases (8 accumulator tiles).  A K stage = 16 input channels: the halo'd x tile (9 x 33 records,
// 19 KB) + all nine taps of the 64 channels (36 KB); two stages in LDS, ONE barrier per stage, the LDS-DMA fetch of stage
// t+1 is issued right after the barrier and runs under the 54 MFMAs per wave of stage t.  The four shifted B fragments of
// a stage are read once; the A fragments of tap i+1 are read under the MFMAs of tap i.
// The previous kernel (conv_f16s_t2v2_kernel<1>: 4 waves, 4 rows x 32 channels, one stage in LDS, DMA -> wait -> 27 MFMAs)
// moved 31 KB per 108 wave-MFMAs; this one moves 55 KB per 432.
//
// Position grid (round 3).  The (H+1) x (W+1) grid on 8 x 32 tiles leaves a tile row / column with ONE valid row or column
// (33 columns = two 32-wide tiles: 57 / 39 / 24 / 13 % of the tiles of the 32² ... 256² layers were such slivers).  The MAIN
// launch now covers the positions (i' < H, j' < W) — exact tiles for the generator's sizes — which produce z rows
// 0..2H-1 and columns 0..2W-1 completely; the last z row (i' = H: taps ky = 2 on x row H-1) and the last z column
// (j' = W: taps kx = 2 on x column W-1) form ONE sequence of H+W+1 edge positions that the EDGE instance of the same
// kernel walks 256 at a time: its x "tile" is the 1-D run S(e) = x[H-1][e] (e <= W), x[e-W-1][W-1] (e > W) and a zero
// record; a tap reads S(e), S(e-1) or zeros.  10 -> 5, 27 -> 17, 85 -> 66, 297 -> 259 workgroup tiles per (image, 64 channels).
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>
#include <cstdlib>
#include <type_traits>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int TB_R = 9, TB_C = 33;                         // rows r0-1 .. r0+7, cols c0-1 .. c0+31 of the image
constexpr int TB_XSLOTS = TB_R * TB_C * 4;                 // 1188 16-byte slots
constexpr int TB_XPIECES = (TB_XSLOTS + 63) / 64;          // 19
constexpr int TB_XBYTES = TB_XPIECES * 1024;
constexpr int TB_WPIECES = 36;
constexpr int TB_PIECES = TB_XPIECES + TB_WPIECES;         // 55 -> 7 per wave (one harmless duplicate)
constexpr int TB_NPW = 7;
constexpr int TB_STAGE = (TB_PIECES + 1) * 1024;           // 57344
constexpr int TB_CST = 2 * TB_STAGE;                       // [64] floats: the out scale of the workgroup's channels
constexpr int TB_SMEM = TB_CST + 64 * 4;

struct T2Big {
    oodgan_conv_args a;
    const uint4* xs;
    SDims xd;
    const float* w_unscale;
    int Hn, Wn, Hout, Wout, tiles_x, tiles_y, mblocks, Mp;      // Hn x Wn = H x W: the positions of the main launch
    long out_plane;
    int etiles;                                                  // EDGE: ceil((H + W + 1) / 256)
    int etotal, eblocks;                                         // edge workgroups, and their count rounded up to a multiple of 8
};

constexpr int TE_RECS = 257;                               // EDGE: records S(e0-1) .. S(e0+255); record 257 = zeros

template <bool EDGE>
__device__ __forceinline__ void t2big_body(const T2Big& p, const uint4* __restrict__ wpk16, unsigned char* smem, int w) {
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    const int mblk = w % p.mblocks;
    w /= p.mblocks;
    const int ntile = EDGE ? p.etiles : p.tiles_x * p.tiles_y;
    const int tile = w % ntile, b = w / ntile;
    const int r0 = (tile / p.tiles_x) * 8, c0 = (tile % p.tiles_x) * 32, m0 = mblk * 64;
    const int e0 = tile * 256;                              // EDGE: first edge position of the tile
    const int M = a.M, H = p.Hn, W = p.Wn;

    // ---- per-lane DMA source offsets (bytes): piece pc = wave + 8*i; x pieces relative to (plane of the chunk), rotation
    // applied; weight pieces relative to the chunk's block
    unsigned off[TB_NPW];
#pragma unroll
    for (int i = 0; i < TB_NPW; ++i) {
        int pc = wave + 8 * i;
        if (pc >= TB_PIECES) pc = TB_PIECES - 1;
        if (pc < TB_XPIECES && EDGE) {
            const int P = pc * 64 + lane, rec = P >> 2, e = e0 - 1 + rec;
            int prow = 0, pcol = 0;                          // padded (0,0): a zero record of the border
            if (rec < TE_RECS && e <= W) { prow = H; pcol = e + 1; }                    // x[H-1][e]   (e = -1, W: zero border)
            else if (rec < TE_RECS && e <= W + H) { prow = e - W; pcol = W; }           // x[e-W-1][W-1]
            off[i] = (unsigned)((((long)prow * p.xd.Wp + pcol) * 4 + (P & 3)) * 16);
        } else if (pc < TB_XPIECES) {
            int P = pc * 64 + lane;
            if (P >= TB_XSLOTS) P = TB_XSLOTS - 1;
            const int row = P / (TB_C * 4), q = P % (TB_C * 4);
            const int c = q >> 2, s = ((q & 3) - ((c >> 2) & 3)) & 3;
            // tile origin = image (r0-1, c0-1) = padded (r0, c0)
            off[i] = (unsigned)((((long)(r0 + row) * p.xd.Wp + (c0 + c)) * 4 + s) * 16);
        } else {
            const int u = (pc - TB_XPIECES) * 64 + lane;    // 16-byte unit inside the 36 x 64 weight block
            off[i] = (unsigned)((((long)(u >> 6) * p.Mp) + m0 + (u & 63)) * 16);
        }
    }
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xs) + (long)b * p.xd.KC * p.xd.plane * 16;
    const long xplane_bytes = p.xd.plane * 16;
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk16);
    const long wchunk_bytes = (long)36 * p.Mp * 16;
    const int nchunk = (a.K + 15) / 16;

    // piece i of this wave for stage t; issued one at a time between the MFMA groups of the previous stage
    auto dma_piece = [&](int t, int buf, int i) {
        int pc = wave + 8 * i;
        if (pc >= TB_PIECES) pc = TB_PIECES - 1;            // wave 7 repeats the last piece (uniform load count)
        const unsigned char* src = (pc < TB_XPIECES ? xb + (long)t * xplane_bytes : wb + (long)t * wchunk_bytes) + off[i];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void*)(smem + buf * TB_STAGE + pc * 1024), 16, 0, 0);
    };
    auto dma_stage = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < TB_NPW; ++i) dma_piece(t, buf, i);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][ph][r] = 0.f;

    // lane-constant fragment offsets: x[i' - da][j' - db] sits at tile (row wave + 1 - da, col l31 + 1 - db)
    unsigned lrd[2][2][2];
#pragma unroll
    for (int da = 0; da < 2; ++da)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int lo = 0; lo < 2; ++lo) {
                if (EDGE) {
                    // edge position e: the last z row (e <= W: i' = H, j' = e; x row H is outside the image, so only the
                    // taps with da = 1 read data: x[H-1][j'-db] = S(e-db)) or the last z column (j' = W, i' = e-W-1: only db = 1:
                    // x[i'-da][W-1] = S(e-da)); S(e) sits in record e-e0+1, record TE_RECS holds zeros
                    const int el = 32 * wave + l31, e = e0 + el;
                    int rec = TE_RECS;
                    if (e <= W) { if (da == 1) rec = el + 1 - db; }
                    else if (e <= W + H) { if (db == 1) rec = el + 1 - da; }
                    lrd[da][db][lo] = rec * 64 + (((half + 2 * lo) & 3) << 4);
                    continue;
                }
                const int c = l31 + 1 - db;
                lrd[da][db][lo] = ((wave + 1 - da) * TB_C + c) * 64 + (((half + 2 * lo + ((c >> 2) & 3)) & 3) << 4);
            }
    const unsigned lwf = (half * 64 + l31) * 16;

    struct AFrag { half8 ah[2], al[2]; };
    auto load_a = [&](AFrag& f, const unsigned char* lw, auto tp_c) {
        constexpr int tp = decltype(tp_c)::value;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f.ah[mt] = *reinterpret_cast<const half8*>(lw + (((tp * 2 + 0) * 2) * 64 + mt * 32) * 16);
            f.al[mt] = *reinterpret_cast<const half8*>(lw + (((tp * 2 + 1) * 2) * 64 + mt * 32) * 16);
        }
    };
    half8 bh[2][2], bl[2][2];
    auto mfma_tap = [&](const AFrag& f, auto tp_c) {
        constexpr int tp = decltype(tp_c)::value;
        constexpr int ky = tp / 3, kx = tp % 3, ph = (ky & 1) * 2 + (kx & 1), da = ky >> 1, db = kx >> 1;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mt], bh[da][db], acc[mt][ph], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[mt], bl[da][db], acc[mt][ph], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][ph] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[mt], bh[da][db], acc[mt][ph], 0, 0, 0);
    };
#define TB_IC(n) std::integral_constant<int, n>{}
#define TB_SB() __builtin_amdgcn_sched_barrier(0)
    // taps are visited in an order that alternates the output phase, so two consecutive taps never hit the same accumulators
    // The epilogue's per-channel scale, requested before the first stage and parked in LDS: loaded where it is used it is a
    // branch, a load and a vmcnt(0) per accumulator register — 32 serialised round trips per tile, each one also waiting for the
    // two stores in front of it.  (Written behind stage 0's request; read after the K loop's barriers.)
    float cst_o = 0.f;
    if (tid < 64 && m0 + tid < M) cst_o = a.out_scale ? a.out_scale[(long)b * a.out_scale_stride + m0 + tid] : 1.f;
    dma_stage(0, 0);
    if (tid < 64) {
        reinterpret_cast<float*>(smem + TB_CST)[tid] = cst_o;
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): written before this wave arrives at the loop's first barrier
    }
    for (int t = 0; t < nchunk; ++t) {
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): only stage t is outstanding
        __builtin_amdgcn_s_barrier();
        const bool pf = t + 1 < nchunk;
        const int nb = (t + 1) & 1;
#define TB_DMA(i) if (pf) dma_piece(t + 1, nb, (i));
        const unsigned char* lx = smem + (t & 1) * TB_STAGE;
        const unsigned char* lw = lx + TB_XBYTES + lwf;
#pragma unroll
        for (int da = 0; da < 2; ++da)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                bh[da][db] = *reinterpret_cast<const half8*>(lx + lrd[da][db][0]);
                bl[da][db] = *reinterpret_cast<const half8*>(lx + lrd[da][db][1]);
            }
        AFrag f0, f1;
        load_a(f0, lw, TB_IC(0));
        TB_SB();
        load_a(f1, lw, TB_IC(1)); TB_DMA(0) TB_SB(); mfma_tap(f0, TB_IC(0)); TB_SB();     // phase 0
        load_a(f0, lw, TB_IC(3)); TB_DMA(1) TB_SB(); mfma_tap(f1, TB_IC(1)); TB_SB();     // phase 1
        load_a(f1, lw, TB_IC(4)); TB_DMA(2) TB_SB(); mfma_tap(f0, TB_IC(3)); TB_SB();     // phase 2
        load_a(f0, lw, TB_IC(2)); TB_DMA(3) TB_SB(); mfma_tap(f1, TB_IC(4)); TB_SB();     // phase 3
        load_a(f1, lw, TB_IC(7)); TB_DMA(4) TB_SB(); mfma_tap(f0, TB_IC(2)); TB_SB();     // phase 0
        load_a(f0, lw, TB_IC(5)); TB_DMA(5) TB_SB(); mfma_tap(f1, TB_IC(7)); TB_SB();     // phase 1
        load_a(f1, lw, TB_IC(6)); TB_DMA(6) TB_SB(); mfma_tap(f0, TB_IC(5)); TB_SB();     // phase 2
        load_a(f0, lw, TB_IC(8)); TB_SB(); mfma_tap(f1, TB_IC(6)); TB_SB();     // phase 0
        mfma_tap(f0, TB_IC(8));                                                 // phase 0 (its predecessor in phase 0 is 6 MFMAs back)
#undef TB_DMA
    }
#undef TB_IC
#undef TB_SB

    // ---- epilogue: z rows 2i', 2i'+1; the two x-phases of a position are one aligned float2
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
    float* yb = a.y + (long)b * M * p.out_plane;
    const float* cst = reinterpret_cast<const float*>(smem + TB_CST);
    if constexpr (EDGE) {
        const int e = e0 + 32 * wave + l31;
        if (e > W + H) return;
        const bool rowt = e <= W;                           // last z row (2H), columns 2e, 2e+1 | last z column (2W), rows 2i', 2i'+1
        const int ie = e - W - 1;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= M) continue;
                const float sc = cst[m - m0] * us;
                float* zp = yb + (long)m * p.out_plane;
                if (rowt) {
                    float* q = zp + (long)(2 * H) * a.out_pitch + 2 * e;
                    q[0] = acc[mt][0][r] * sc;
                    if (e < W) q[1] = acc[mt][1][r] * sc;
                } else {
                    zp[(long)(2 * ie) * a.out_pitch + 2 * W] = acc[mt][0][r] * sc;
                    zp[(long)(2 * ie + 1) * a.out_pitch + 2 * W] = acc[mt][2][r] * sc;
                }
            }
        return;
    }
    const int ip = r0 + wave, jp = c0 + l31;
    const int zx = 2 * jp;
    if (ip >= H || jp >= W) return;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= M) continue;
            const float sc = cst[m - m0] * us;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int zy = 2 * ip + py;
                if (zy >= p.Hout) continue;
                // out_pitch and out_plane are even (host guarantees): 8-byte aligned
                *reinterpret_cast<float2*>(yb + (long)m * p.out_plane + (long)zy * a.out_pitch + zx) =
                    make_float2(acc[mt][py * 2 + 0][r] * sc, acc[mt][py * 2 + 1][r] * sc);
            }
        }
}


// ONE launch: the first p.etotal (rounded up to a multiple of 8) workgroups walk the edge positions, the others the main tiles.
// The edge workgroups are few (24-64) and latency-bound — a full K loop for a single tile, 60-70 us — so as a launch of their
// own they cost most of what the exact main grid saves; dispatched first inside the main launch they run beside its tiles.
__global__ __launch_bounds__(512) void conv_f16s_t2big_kernel(const T2Big p, const uint4* __restrict__ wpk16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bid = blockIdx.x;
    if (bid < p.eblocks) {
        if (bid < p.etotal) t2big_body<true>(p, wpk16, smem, bid);
        return;
    }
    t2big_body<false>(p, wpk16, smem, xcd_remap(bid - p.eblocks, gridDim.x - p.eblocks));
}

}  // namespace

namespace oodgan {

bool t2_big_eligible(const oodgan_conv_args& a) {
    if (!(a.mode == OODGAN_CONV_T2 && a.x_sform && a.M >= 64 && a.y != nullptr && a.in_scale == nullptr && a.in_shift == nullptr &&
          a.dotx == nullptr && a.noise == nullptr && a.bias == nullptr && a.act == OODGAN_ACT_NONE && a.ys == nullptr))
        return false;
    // enough (8x32 positions x 64 channels) items to fill the chip; smaller layers keep the latency-oriented instance
    const long items = (long)((a.Hin + 7) / 8) * ((a.Win + 31) / 32) * a.B * ((a.M + 63) / 64);
    // 128 work items = half the CUs: what a sub-batch of 2-3 images (three concurrent streams) brings to the 64² / 32² layers.  Whole loop,
    // 3 streams, same box: threshold 256 -> 5.67 img/s, 128 -> 5.79 (one stream, batch 8: the 32² input gradient 169 -> 150 us)
    return items >= tunable(OODGAN_TUN_T2_BIG_MIN_ITEMS);      // default 128; tests lower it to reach this kernel with small tensors
}

int launch_t2_big(const oodgan_conv_args& a_in, const void* wpk16, const float* unscale, hipStream_t st) {
    T2Big p;
    p.a = a_in;
    oodgan_conv_args& a = p.a;
    p.Hn = a.Hin; p.Wn = a.Win; p.Hout = 2 * a.Hin + 1; p.Wout = 2 * a.Win + 1;
    if (a.out_pitch == 0) a.out_pitch = p.Wout + 1;
    OODGAN_REQUIRE((a.out_pitch & 1) == 0 && a.out_pitch >= p.Wout + 1, "conv3x3 T2: out_pitch must be even and > 2W+1 (got %d)", a.out_pitch);
    p.out_plane = (long)p.Hout * a.out_pitch;
    p.xs = reinterpret_cast<const uint4*>(a.x);
    p.xd = sform_dims(a.K, a.Hin, a.Win);
    p.w_unscale = unscale;
    p.tiles_y = (p.Hn + 7) / 8;
    p.tiles_x = (p.Wn + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    p.mblocks = (a.M + 63) / 64;
    OODGAN_REQUIRE(p.xd.plane * 16 < (1L << 32), "conv3x3 T2 big: input plane too large");
    p.etiles = (a.Hin + a.Win + 1 + 255) / 256;
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks, etotal = (long)p.etiles * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_t2big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TB_SMEM), true);
    (void)once;
    static_assert(TE_RECS * 4 + 4 <= TB_XPIECES * 64, "edge records + the zero record fit the x region");
    p.etotal = (int)etotal;
    p.eblocks = (int)((etotal + 7) / 8 * 8);             // keeps the main tiles' blockIdx -> XCD chunk mapping intact
    OODGAN_REQUIRE(total + p.eblocks < (1L << 31), "conv3x3: grid too large");
    hipLaunchKernelGGL(conv_f16s_t2big_kernel, dim3((unsigned)(total + p.eblocks)), dim3(512), TB_SMEM, st, p, reinterpret_cast<const uint4*>(wpk16));
    return check_launch("conv3x3_f16s_t2big");
}

}  // namespace oodgan
