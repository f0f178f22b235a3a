// Split-f16 3x3 stride-1 conv for the >= 64-channel layers (64² ... 512² images): the matrix-bound middle of the
// generator (reference src/ops/StyleGAN/model.py:233-274, forward and input gradient).  Same arithmetic as
// conv_f16s_s1v2_kernel; different pipeline.  The v2 kernel keeps two 64 KB stages in LDS as two anti-phase groups and
// is bound by the LDS-DMA latency of one stage per step.  Here ONE workgroup of 8 waves owns a 16 x 32 pixel tile (twice
// the pixels per weight byte): a K stage is 39 KB of x + 36 KB of weights feeding 224 MFMAs (16x16x32) in each of the 8
// waves, two stages live in LDS, and the fetch of stage t+1 runs under the MFMAs of stage t; both waves of a SIMD issue
// MFMAs, so one covers the other's fragment-read latency.
//   * x tile 18 x 34 records of 64 B, the four 16-byte slots of record c rotated by 2*((c>>2)&1) through the DMA source
//     address: with this kernel's lane layout (lane = pixel + 16 * K group) every one of ds_read_b128's four 16-lane groups
//     ({0-3,12-15,20-27}, ...) then covers the 16 slots of a bank row once, for every tap shift (the (c>>2)&3 rotation of the
//     32x32x16 kernels gave 2-way conflicts on a quarter of the slots here: 32 % of the LDS-active cycles), no padding;  weights in the packed order [tap][hi|lo][k-half][64][8] (36 KB per 16 input channels);
//   * register epilogue: fp32 runs of 64 contiguous bytes per lane row, fused out-scale / noise / bias / lrelu, or
//     the style-gradient dot (backward) reduced through DPP + LDS.
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>
#include <cstdlib>
#include <type_traits>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int BG_R = 18, BG_C = 34;                        // halo'd 16 x 32 tile
constexpr int BG_XSLOTS = BG_R * BG_C * 4;                 // 2448 16-byte slots
constexpr int BG_XPIECES = (BG_XSLOTS + 63) / 64;          // 39
constexpr int BG_XBYTES = BG_XPIECES * 1024;               // 39936
constexpr int BG_WPIECES = 36;                             // 36 rows x 64 channels x 16 B
constexpr int BG_STAGE = BG_XBYTES + BG_WPIECES * 1024;    // 76800
constexpr int BG_PIECES = BG_XPIECES + BG_WPIECES;         // 75
constexpr int BG_RED = 2 * BG_STAGE;                       // 4 waves x 64 floats
constexpr int BG_SMEM = BG_RED + 8 * 64 * 4;

struct BigConv {
    oodgan_conv_args a;
    const uint4* xs;
    SDims xd;
    const float* w_unscale;
    int tiles_x, tiles_y, mblocks, Mp;
    long out_plane;
    SDims yd;                // YS instance: S-form of the (M, H, W) output
};

// The matrix instruction is v_mfma_f32_16x16x32_f16: under a dense MFMA stream the chip holds a higher clock with this shape
// than with 32x32x16 (MI355X_MICROARCH.md, DVFS item 7; tools/mfma_shape_probe.hip: this kernel's bare tap loop 10 % faster;
// the whole kernel, against its 32x32x16 predecessor on the same box: -2.6 % over the four layers, -5 % on the input-gradient
// instances).  K = 32 of one instruction = the 16 channels of TWO TAPS: A = [w(tap a) | w(tap b)] (16 channels x 32), B = [x shifted by tap a ; by tap b]
// (32 x 16 pixels), so the hi*hi, hi*lo and lo*hi products of a tap pair are three instructions per 16 x 16 tile and the
// fragment reads per tap stay at 8.  The ninth tap of a chunk takes the hi/lo form instead: [w_hi|w_hi] x [x_hi;x_lo] gives
// hi*hi + hi*lo, [w_lo|w_lo] x [x_hi;0] gives lo*hi (zeros from an LDS region).  A wave owns 64 channels x 2 rows x 32 pixels =
// 4 x 4 tiles of 16 x 16 (64 accumulator registers).  Two stages in LDS, ONE barrier per stage: after it every wave has
// finished stage t-1 (its buffer is free: the LDS-DMA pieces of stage t+1 are issued one at a time BETWEEN the MFMA groups of
// stage t — a global_load_lds costs the issuing wave ~60-180 cycles (MI355X_MICROARCH.md), and a burst of ten right after the
// barrier stalls every wave of the CU at once) and stage t has landed (each wave waited for its own loads).
// PRE (input-gradient instance, oodgan_conv_args.dot_actgrad): the result is the gradient w.r.t. dotx, the OUTPUT of the
// up-sampling StyledConv below; the epilogue applies that layer's FusedLeakyReLU backward (fused_act.py:25-58) to the values
// it already holds — y <- dx * (dotx > 0 ? sqrt2 : 0.2 sqrt2) — so the blur^T producer that follows reads one tensor
// instead of two.  Three VALU operations per value: the kernel's waves have no spare issue slots for more (the sums of
// that layer's demodulation gradient are finished by the producer and the dot partials, include/oodgan.h).
constexpr int BG_ZERO = BG_SMEM;                           // 3.5 KB of zeros (the B operand's second half of the lo*hi product of tap 8)
constexpr int BG_ZBYTES = 3584;
constexpr int BG_CST = BG_SMEM + BG_ZBYTES;                // [3][64] floats: out scale, bias, PReLU slope of the workgroup's channels
constexpr int BG_NZ = BG_CST + 3 * 64 * 4;                 // forward: the tile's 16 x 32 noise values, by LDS-DMA before the K loop (2 KB)
constexpr int BG_RGBW = BG_NZ + 2048;                      // YS instance with the fused ToRGB partial: [3 colours][64 channels] rgb_scale * w_rgb * s_rgb
constexpr int BG_SMEM16 = BG_RGBW + 3 * 64 * 4;            // 162816 of 163840
typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifdef OODGAN_CLOCK_STAMP
// Diagnostic build only (make STAMP=1 -> liboodgan_hip_stamp.so; MI355X_MICROARCH.md, DVFS item 6): the in-kernel clock is
// d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped once around the K loop.  The stamps go to a buffer of their own that no
// kernel reads; the production library contains none of this.
__device__ unsigned long long* g_s1big_stamp = nullptr;
__device__ long g_s1big_stamp_n = 0;
#endif

// YS (forward only, oodgan_conv_args.ys): the activated output x ys_scale[b,m] is ALSO (y != NULL) or ONLY (y == NULL) written as the
// S-form input of the next conv — lane (n16, g) holds channels 4g .. 4g+3 of a 16-channel block of its pixel: 8 bytes of the hi slot
// and 8 bytes of the lo slot of that pixel's record, 16 lanes = 16 consecutive records (SAMM's AlignNet: conv -> PReLU -> conv
// without the fp32 tensor, its range measurement and its conversion pass in between)
// G2 (input-gradient instances, oodgan_conv_args.x_hi_only): the lo half of the S-form operand is dropped — x_hi * (w_hi + w_lo), TWO matrix
// instructions per tap pair instead of three, and ONE for the ninth tap ([w_hi|w_lo] x [x_hi;x_hi]): 9 instead of 14 per 16 x 16 tile and
// chunk.  The back-propagated gradient is rounded to f16 (2^-11 relative, zero-mean, independent per element and step); the weights keep
// their 22 bits, so no systematic error enters (precision 'f16s-g2').
template <bool DOT, bool PRE, bool YS = false, bool G2 = false>
__global__ __launch_bounds__(512) void conv_f16s_s1big_kernel(const BigConv p, const uint4* __restrict__ wpk16) {
    constexpr int NW = 8, NT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;

    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int mblk = w % p.mblocks;
    w /= p.mblocks;
    const int ntile = p.tiles_x * p.tiles_y;
    const int tile = w % ntile, b = w / ntile;
    const int ty = tile / p.tiles_x, tx = tile % p.tiles_x;
    const int r0 = ty * 16, c0 = tx * 32, m0 = mblk * 64;
    const int H = a.Hin, W = a.Win, M = a.M;

    constexpr int NPW = (BG_PIECES + NW - 1) / NW;          // 10
    unsigned off[NPW];
    // XH (G2 instances with oodgan_conv_args.x_hi_only == 2): x holds 32-byte hi-only records (oodgan_actbwd_fuse.ys_hi_only) at half the in-plane offsets;
    // the LDS image keeps its 64-byte records (this instance never reads the lo slots), the lanes of a piece that would carry a lo slot stay off
    const bool xh = G2 && a.x_hi_only == 2;
    unsigned xmask = 0;      // bit i: this lane takes part in piece i
    const int KC = p.xd.KC;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int pc = wave + NW * i;
        if (pc < BG_XPIECES) {
            int P = pc * 64 + lane;
            if (P >= BG_XSLOTS) P = BG_XSLOTS - 1;
            const int row = P / (BG_C * 4), q = P % (BG_C * 4);
            const int c = q >> 2, sl = ((q & 3) - 2 * ((c >> 2) & 1)) & 3;
            const int rr = min(r0 + row, p.xd.Hp - 1);
            if (xh) {
                off[i] = (unsigned)((((long)rr * p.xd.Wp + (c0 + c)) * 2 + (sl & 1)) * 16);
                if (sl < 2) xmask |= 1u << i;
            } else {
                off[i] = (unsigned)((((long)rr * p.xd.Wp + (c0 + c)) * 4 + sl) * 16);
                xmask |= 1u << i;
            }
        } else {
            xmask |= 1u << i;
            const int u = (pc - BG_XPIECES) * 64 + lane;
            const int row = u >> 6, j = u & 63;
            off[i] = (unsigned)((((long)row * p.Mp) + m0 + j) * 16);
        }
    }
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xs) + (long)b * KC * p.xd.plane * 16;
    const long xplane_bytes = p.xd.plane * 16;
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk16);
    const long wchunk_bytes = (long)36 * p.Mp * 16;
    const int nchunk = (a.K + 15) / 16;
    auto dma_piece = [&](int t, int buf, int i) {
        const int pc = wave + NW * i;
        if (pc >= BG_PIECES) return;
        unsigned char* dst = smem + buf * BG_STAGE;
        const unsigned char* src = (pc < BG_XPIECES ? xb + (long)t * xplane_bytes : wb + (long)t * wchunk_bytes) + off[i];
        if (!G2 || ((xmask >> i) & 1u))
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void*)(dst + pc * 1024), 16, 0, 0);
    };
    for (int i = tid; i < BG_ZBYTES / 16; i += 512) reinterpret_cast<uint4*>(smem + BG_ZERO)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();         // the loop's barriers wait for vmcnt only: make the zero region visible to every wave here

    f32x4v acc[4][4];        // [M-tile of 16 channels][row nt * 2 + column half]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[mt][n] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // ---- lane-constant fragment offsets.  Lane = (row / column 16 index n16, K group g): K groups 0,1 = channels 0-7 / 8-15 of
    // the pair's first tap, 2,3 = of its second tap.  Pairs are consecutive taps (2p, 2p+1), tap = ky*3 + kx.
    const unsigned laneA = (unsigned)((((g & 1) * 64 + n16) * 16) + (g >> 1) * 4096);       // + 2p*4096 + hl*2048 + mt*256
    const unsigned laneA1 = (unsigned)(((g & 1) * 64 + n16) * 16);                             // ninth tap: both halves the same tap
    unsigned offBh[4], offBl[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
        const int tp = 2 * pp + (g >> 1), ky = tp / 3, kx = tp % 3, col = n16 + kx;
        offBh[pp] = (unsigned)(ky * (BG_C * 64) + col * 64 + ((((g & 1) + 2 * ((col >> 2) & 1)) & 3) << 4));
        offBl[pp] = (unsigned)(ky * (BG_C * 64) + col * 64 + ((((g & 1) + 2 + 2 * ((col >> 2) & 1)) & 3) << 4));
    }
    const int col8 = n16 + 2;
    const unsigned offB1 = (unsigned)(2 * (BG_C * 64) + col8 * 64 + ((((G2 ? (g & 1) : g) + 2 * ((col8 >> 2) & 1)) & 3) << 4));      // [x_hi ; x_lo] of tap 8: slot g (G2: [x_hi ; x_hi])

    struct AF { half8 x[2], y[2]; };         // pair: w_hi, w_lo of two M-tiles; ninth tap: [w_hi|w_hi], [w_lo|w_lo]
    struct BF { half8 u[4], v[4]; };         // pair: x_hi, x_lo of the four N-tiles; ninth tap: [x_hi;x_lo], [x_hi;0]
    auto load_a = [&](AF& f, const unsigned char* lw, auto pp_c, auto mh_c) {
        constexpr int pp = decltype(pp_c)::value, mh = decltype(mh_c)::value;
        // G2, ninth tap: A = [w_hi | w_lo] — K groups 0,1 read the hi rows, 2,3 the lo rows of the tap
        const unsigned char* q = lw + (pp < 4 ? laneA + 2 * pp * 4096 : laneA1 + 8 * 4096 + (G2 ? (g >> 1) * 2048 : 0));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f.x[i] = *reinterpret_cast<const half8*>(q + (2 * mh + i) * 256);
            if (!(G2 && pp == 4)) f.y[i] = *reinterpret_cast<const half8*>(q + 2048 + (2 * mh + i) * 256);
        }
    };
    auto load_b = [&](BF& f, const unsigned char* lx, const unsigned char* lz, auto pp_c, auto part_c) {       // part 0: u, 1: v
        constexpr int pp = decltype(pp_c)::value, part = decltype(part_c)::value;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int o = (n >> 1) * (BG_C * 64) + (n & 1) * 1024;
            if (G2) {
                // hi halves only; ninth tap: B = [x_hi ; x_hi] (every K group reads the hi slot of its k-half)
                if (part == 0) f.u[n] = *reinterpret_cast<const half8*>(lx + (pp < 4 ? offBh[pp < 4 ? pp : 0] : offB1) + o);
            } else if (pp < 4) {
                if (part == 0) f.u[n] = *reinterpret_cast<const half8*>(lx + offBh[pp < 4 ? pp : 0] + o);
                else f.v[n] = *reinterpret_cast<const half8*>(lx + offBl[pp < 4 ? pp : 0] + o);
            } else {
                if (part == 0) f.u[n] = *reinterpret_cast<const half8*>(lx + offB1 + o);
                else f.v[n] = *reinterpret_cast<const half8*>(lz + o);         // K groups 0,1: x_hi of tap 8; groups 2,3: zeros
            }
        }
    };
    auto mfma_half = [&](const AF& fa, const BF& fb, auto pp_c, auto mh_c) {
        constexpr int pp = decltype(pp_c)::value, mh = decltype(mh_c)::value;
        if (G2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[2 * mh + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.x[i], fb.u[n], acc[2 * mh + i][n], 0, 0, 0);
            if (pp < 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[2 * mh + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.y[i], fb.u[n], acc[2 * mh + i][n], 0, 0, 0);
            }
        } else if (pp < 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[2 * mh + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.x[i], fb.u[n], acc[2 * mh + i][n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[2 * mh + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.x[i], fb.v[n], acc[2 * mh + i][n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[2 * mh + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.y[i], fb.u[n], acc[2 * mh + i][n], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[2 * mh + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.x[i], fb.u[n], acc[2 * mh + i][n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[2 * mh + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.y[i], fb.v[n], acc[2 * mh + i][n], 0, 0, 0);
        }
    };
#define BH_IC(n) std::integral_constant<int, n>{}
#define BH_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef OODGAN_CLOCK_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Per-channel constants of the epilogue, requested before the first stage and parked in LDS.  Loaded where they are used
    // (`m < M ? out_scale[m] : 0` per accumulator register) each one is a branch around a load followed by vmcnt(0): sixteen
    // serialised round trips per tile in front of the stores.  The ds_write waits for stage 0 like the loop's first barrier does;
    // the epilogue reads the table behind the loop's barriers.
    float cst_o = 0.f, cst_b = 0.f, cst_s = 1.f, rgb_c0 = 0.f, rgb_c1 = 0.f, rgb_c2 = 0.f;
    if (tid < 64 && m0 + tid < M) {
        const int m = m0 + tid;
        cst_o = a.out_scale ? a.out_scale[(long)b * a.out_scale_stride + m] : 1.f;
        if (!DOT && a.bias) cst_b = a.bias[m];
        if (!DOT && a.act == OODGAN_ACT_PRELU) cst_s = a.slope[m];
        if (YS && a.rgb_y) {                            // fused ToRGB partial of this 64-channel block (oodgan_conv_args.rgb_*)
            const float sv = a.rgb_scale * a.rgb_s[(long)b * a.rgb_s_stride + m];
            rgb_c0 = sv * a.rgb_w[m]; rgb_c1 = sv * a.rgb_w[M + m]; rgb_c2 = sv * a.rgb_w[2 * M + m];
        }
    }
    // Forward: the noise of this wave's two rows, one dword per lane (lanes 0-31 row 0, 32-63 row 1) from clamped pixels, straight into
    // LDS with the first stage.  In the epilogue `if (ok) nz = noise[...]` was a branch around a load followed by its use, once per
    // N-tile: four serialised round trips per tile in front of the stores (round 4: 620 -> see LABNOTES.md 13.6 for the 512² layer).
    const bool has_nz = !DOT && a.noise != nullptr;
    if (has_nz) {
        const int ry = min(r0 + wave * NT + (lane >> 5), H - 1), rx = min(c0 + (lane & 31), W - 1);
        const float* src = a.noise + (long)(a.noise_batch > 1 ? b : 0) * H * W + (long)ry * W + rx;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void*)(smem + BG_NZ + wave * 256), 4, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) dma_piece(0, 0, i);
    if (tid < 64) {
        float* cst = reinterpret_cast<float*>(smem + BG_CST);
        cst[tid] = cst_o;
        if (!DOT) { cst[64 + tid] = cst_b; cst[128 + tid] = cst_s; }
        if (YS) {
            float* rw = reinterpret_cast<float*>(smem + BG_RGBW);
            rw[tid] = rgb_c0; rw[64 + tid] = rgb_c1; rw[128 + tid] = rgb_c2;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): written before this wave arrives at the loop's first barrier
    }
    for (int t = 0; t < nchunk; ++t) {
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): only stage t is outstanding
        __builtin_amdgcn_s_barrier();
        const bool pf = t + 1 < nchunk;
        const int nb = (t + 1) & 1;
        const unsigned char* lx = smem + (t & 1) * BG_STAGE + (wave * NT) * (BG_C * 64);
        const unsigned char* lw = smem + (t & 1) * BG_STAGE + BG_XBYTES;
        // [x_hi ; 0] of tap 8: K groups 0,1 read the hi slots of the record, groups 2,3 the zero region (same immediate offsets)
        const unsigned col8r = (unsigned)(2 * (BG_C * 64) + col8 * 64 + ((((g & 1) + 2 * ((col8 >> 2) & 1)) & 3) << 4));
        const unsigned char* lz = g < 2 ? lx + col8r : smem + BG_ZERO;
        AF a0, a1;
        BF b0, b1;
#define BH_DMA(i) if (pf && (i) < NPW) dma_piece(t + 1, nb, (i));
        // ten half steps: pairs 0..3 and the ninth tap (index 4), each for the M-tile halves 0 and 1; the A fragments of the next
        // half step and half of the next pair's B fragments are read under the MFMAs of the current one
        load_b(b0, lx, lz, BH_IC(0), BH_IC(0)); load_b(b0, lx, lz, BH_IC(0), BH_IC(1));
        load_a(a0, lw, BH_IC(0), BH_IC(0));
        BH_SB();
        load_a(a1, lw, BH_IC(0), BH_IC(1)); load_b(b1, lx, lz, BH_IC(1), BH_IC(0)); BH_DMA(0) BH_SB(); mfma_half(a0, b0, BH_IC(0), BH_IC(0)); BH_SB();
        load_a(a0, lw, BH_IC(1), BH_IC(0)); load_b(b1, lx, lz, BH_IC(1), BH_IC(1)); BH_DMA(1) BH_SB(); mfma_half(a1, b0, BH_IC(0), BH_IC(1)); BH_SB();
        load_a(a1, lw, BH_IC(1), BH_IC(1)); load_b(b0, lx, lz, BH_IC(2), BH_IC(0)); BH_DMA(2) BH_SB(); mfma_half(a0, b1, BH_IC(1), BH_IC(0)); BH_SB();
        load_a(a0, lw, BH_IC(2), BH_IC(0)); load_b(b0, lx, lz, BH_IC(2), BH_IC(1)); BH_DMA(3) BH_SB(); mfma_half(a1, b1, BH_IC(1), BH_IC(1)); BH_SB();
        load_a(a1, lw, BH_IC(2), BH_IC(1)); load_b(b1, lx, lz, BH_IC(3), BH_IC(0)); BH_DMA(4) BH_SB(); mfma_half(a0, b0, BH_IC(2), BH_IC(0)); BH_SB();
        load_a(a0, lw, BH_IC(3), BH_IC(0)); load_b(b1, lx, lz, BH_IC(3), BH_IC(1)); BH_DMA(5) BH_SB(); mfma_half(a1, b0, BH_IC(2), BH_IC(1)); BH_SB();
        load_a(a1, lw, BH_IC(3), BH_IC(1)); load_b(b0, lx, lz, BH_IC(4), BH_IC(0)); BH_DMA(6) BH_SB(); mfma_half(a0, b1, BH_IC(3), BH_IC(0)); BH_SB();
        load_a(a0, lw, BH_IC(4), BH_IC(0)); load_b(b0, lx, lz, BH_IC(4), BH_IC(1)); BH_DMA(7) BH_SB(); mfma_half(a1, b1, BH_IC(3), BH_IC(1)); BH_SB();
        load_a(a1, lw, BH_IC(4), BH_IC(1)); BH_DMA(8) BH_SB(); mfma_half(a0, b0, BH_IC(4), BH_IC(0)); BH_SB();
        BH_DMA(9) BH_SB();
        mfma_half(a1, b0, BH_IC(4), BH_IC(1));
#undef BH_DMA
    }
#undef BH_IC
#undef BH_SB
#ifdef OODGAN_CLOCK_STAMP
    if (tid == 0 && g_s1big_stamp && (long)blockIdx.x < g_s1big_stamp_n) {
        g_s1big_stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_t0;
        g_s1big_stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif

    // ---- epilogue from the accumulators: lane (n16, g) holds, per M-tile mt and N-tile n, channels 16*mt + 4*g + r (r = 0..3) of
    // pixel (row nt = n >> 1, column 16*(n & 1) + n16): 64-byte runs per channel and store
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
    const float nw = (!DOT && a.noise) ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    float* red = reinterpret_cast<float*>(smem + BG_RED);
    float osc[16], bia[16], slp[16];
    unsigned moff[16], doff[16];
    float dsum[16];
    const bool prelu = !DOT && a.act == OODGAN_ACT_PRELU;
    const float* cst = reinterpret_cast<const float*>(smem + BG_CST);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int j = 16 * (q >> 2) + 4 * g + (q & 3), m = m0 + j;
        const bool mok = m < M;
        osc[q] = cst[j] * us;
        bia[q] = DOT ? 0.f : cst[64 + j];
        slp[q] = DOT ? 1.f : cst[128 + j];
        moff[q] = mok ? (unsigned)((long)m * p.out_plane * 4) : 0xFFFFFFFFu;
        doff[q] = mok ? (unsigned)((long)m * H * W * 4) : 0u;
        dsum[q] = 0.f;
    }
    const bool mfull = m0 + 64 <= M;
    float ysc[16], ysmax = 0.f;
    if (YS) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + 16 * (q >> 2) + 4 * g + (q & 3);
            ysc[q] = a.ys_scale ? a.ys_scale[(long)b * a.ys_scale_stride + min(m, M - 1)] : 1.f;
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int py = r0 + wave * NT + (n >> 1), px = c0 + 16 * (n & 1) + n16;
        const bool ok = py < H && px < W;
        float nz = 0.f;
        if (has_nz) nz = nw * reinterpret_cast<const float*>(smem + BG_NZ)[wave * 64 + (n >> 1) * 32 + 16 * (n & 1) + n16];
        unsigned char* yr = reinterpret_cast<unsigned char*>(a.y) + ((long)b * M * p.out_plane + (long)py * a.out_pitch + px) * 4;
        const unsigned char* dr = DOT ? reinterpret_cast<const unsigned char*>(a.dotx) + ((long)b * M * H * W + (long)py * W + px) * 4 : nullptr;
        float o[16], dv[16];
        if (DOT) {
#pragma unroll
            for (int q = 0; q < 16; ++q) dv[q] = 0.f;
            if (ok) {
                if (mfull) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) dv[q] = *reinterpret_cast<const float*>(dr + doff[q]);
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (moff[q] != 0xFFFFFFFFu) dv[q] = *reinterpret_cast<const float*>(dr + doff[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float v = acc[q >> 2][n][q & 3];
            o[q] = v * osc[q];
            if (DOT) {
                dsum[q] += (v * us) * dv[q];
                if (PRE) o[q] *= dv[q] > 0.f ? kSqrt2 : 0.2f * kSqrt2;
            } else {
                o[q] += nz + bia[q];
                if (a.act == OODGAN_ACT_LRELU) o[q] = (o[q] > 0.f ? o[q] : 0.2f * o[q]) * kSqrt2;
                else if (prelu) o[q] = o[q] > 0.f ? o[q] : slp[q] * o[q];
            }
        }
        if (ok && (!YS || a.y != nullptr)) {
            if (mfull) {
#pragma unroll
                for (int q = 0; q < 16; ++q) *reinterpret_cast<float*>(yr + moff[q]) = o[q];
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (moff[q] != 0xFFFFFFFFu) *reinterpret_cast<float*>(yr + moff[q]) = o[q];
            }
        }
        if (YS && !DOT && ok) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (m0 + 16 * mt < M) {          // M % 16 == 0 (s1_big_eligible)
                    const float v0 = o[4 * mt] * ysc[4 * mt], v1 = o[4 * mt + 1] * ysc[4 * mt + 1], v2 = o[4 * mt + 2] * ysc[4 * mt + 2],
                                v3 = o[4 * mt + 3] * ysc[4 * mt + 3];
                    ysmax = fmaxf(fmaxf(ysmax, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
                    uint2 hi, lo;
                    split_pair(v0, v1, hi.x, lo.x);
                    split_pair(v2, v3, hi.y, lo.y);
                    unsigned char* rec = reinterpret_cast<unsigned char*>(a.ys) + sform_unit(p.yd, b, (m0 >> 4) + mt, py, px, g >> 1) * 16 + (g & 1) * 8;
                    *reinterpret_cast<uint2*>(rec) = hi;
                    *reinterpret_cast<uint2*>(rec + 32) = lo;
                }
            }
        }
        if (YS && !DOT && a.rgb_y) {
            // ToRGB partial of this pixel over the workgroup's 64 channels: 16 per lane, then the four K-group lanes of the pixel
            const float* rw = reinterpret_cast<const float*>(smem + BG_RGBW);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = 16 * (q >> 2) + 4 * g + (q & 3);
                c0 += rw[j] * o[q]; c1 += rw[64 + j] * o[q]; c2 += rw[128 + j] * o[q];
            }
            c0 += __shfl_xor(c0, 16, 64); c1 += __shfl_xor(c1, 16, 64); c2 += __shfl_xor(c2, 16, 64);
            c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64);
            if (ok && g == 0) {
                float* rp = a.rgb_y + (((long)mblk * a.B + b) * 3) * ((long)H * W) + (long)py * W + px;
                rp[0] = c0; rp[(long)H * W] = c1; rp[2 * (long)H * W] = c2;
            }
        }
    }
    if (YS && a.ys_vmax) record_vmax(a.ys_vmax, b, ysmax);      // forward range control of the conv that reads `ys` (sform.hpp)
    if (DOT) {
        // sum over the 16 pixels of a lane row (DPP, the total in every lane of the row), then across the waves through LDS
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float v = dsum[q];
            v += dpp_take<0xB1, 0xF>(v);
            v += dpp_take<0x4E, 0xF>(v);
            v += dpp_take<0x141, 0xF>(v);
            v += dpp_take<0x140, 0xF>(v);
            if (n16 == 0) red[wave * 64 + 16 * (q >> 2) + 4 * g + (q & 3)] = v;
        }
        __syncthreads();
        if (tid < 64 && m0 + tid < M) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) v += red[wv * 64 + tid];
            float* dp = a.dot_part + ((long)b * M + m0 + tid) * a.dot_nparts;
            const int tx8 = (W + 31) / 32;
            dp[(long)(2 * ty) * tx8 + tx] = v;
            if (2 * ty + 1 < (H + 7) / 8) dp[(long)(2 * ty + 1) * tx8 + tx] = 0.f;
        }
    }
}

}  // namespace

namespace oodgan {

bool s1_big_eligible(const oodgan_conv_args& a) {
    // Measured (tools/bench_conv.py, B=8, us per launch; v2 two-group tile kernel -> this kernel with 8 waves):
    // 512->512 @64²: 455 -> 393 (394 TF/s), 256->256 @128²: 435 -> 407, 128->128 @256²: 484 -> 464.  With 4 waves (one
    // per SIMD) it is slower than v2 (474 / 475 / 552): ablating its DMA does not change the time — the MFMA +
    // fragment-read stream of a single wave per SIMD runs at 31 ns per MFMA (a bare MFMA loop: 17-20 ns), two waves per
    // SIMD overlap each other's LDS waits.
    // In the inversion loop (rocprof, per launch): forward 397 us, input gradient with the dot epilogue 443 us, against
    // ~458 us for either v2 instance.  (The dot epilogue first cost 561 us: one conditional load per value serialised
    // 64 memory latencies; the loads of a row are now issued together.)
    const int min_k = 64;                // 64 -> 64 channels @512²: 589 -> 491 us forward
    if (!(a.mode == OODGAN_CONV_S1 && a.x_sform && a.K >= min_k && a.M >= 64 && (a.ys == nullptr ? (a.y != nullptr && a.rgb_y == nullptr) : (a.dotx == nullptr && a.M % 16 == 0)) &&
          (a.act == OODGAN_ACT_NONE || a.act == OODGAN_ACT_LRELU || (a.act == OODGAN_ACT_PRELU && a.slope)) && a.in_scale == nullptr &&
          a.in_shift == nullptr &&
          !(a.dotx && (a.noise || a.bias || a.act != OODGAN_ACT_NONE))))
        return false;
    // enough 16x32 tiles to fill the chip; smaller layers keep the latency-oriented instances
    const long items = (long)((a.Hin + 15) / 16) * ((a.Win + 31) / 32) * a.B * ((a.M + 63) / 64);
    // 128 work items = half the CUs: what a sub-batch of 2-3 images (three concurrent streams) brings to the 64² / 32² layers.  Whole loop,
    // 3 streams, same box: threshold 256 -> 5.67 img/s, 128 -> 5.79 (one stream, batch 8: the 32² input gradient 169 -> 150 us)
    return items >= tunable(OODGAN_TUN_S1_BIG_MIN_ITEMS);      // default 128; tests lower it to reach this kernel with small tensors
}

int launch_s1_big(const oodgan_conv_args& a_in, const void* wpk16, const float* unscale, hipStream_t st) {
    BigConv p;
    p.a = a_in;
    oodgan_conv_args& a = p.a;
    if (a.out_pitch == 0) a.out_pitch = a.Win;
    p.out_plane = (long)a.Hin * a.out_pitch;
    p.xs = reinterpret_cast<const uint4*>(a.x);
    p.xd = sform_dims(a.K, a.Hin, a.Win);
    p.w_unscale = unscale;
    p.tiles_y = (a.Hin + 15) / 16;
    p.tiles_x = (a.Win + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    p.mblocks = (a.M + 63) / 64;
    p.yd = sform_dims(a.M, a.Hin, a.Win);
    if (a.dotx) {
        OODGAN_REQUIRE(a.dot_part != nullptr, "conv3x3: dotx without dot_part");
        OODGAN_REQUIRE(a.dot_nparts == ((a.Hin + 7) / 8) * p.tiles_x, "conv3x3 f16s S1: dot_nparts %d != %d", a.dot_nparts,
                       ((a.Hin + 7) / 8) * p.tiles_x);
    }
    OODGAN_REQUIRE((long)a.M * p.out_plane * 4 < (1L << 32) && (long)a.M * a.Hin * a.Win * 4 < (1L << 32), "conv3x3 big: plane too large");
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM16),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM16),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM16),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM16),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM16),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM16), true);
    (void)once;
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
    // 8 waves: two per SIMD keep the MFMA pipe fed while the partner waits on LDS (a 4-wave variant was slower than the
    // v2 tile kernel: 474 / 475 / 552 us against 393 / 407 / 464 on the 64² / 128² / 256² layers)
    OODGAN_REQUIRE(!a.dot_actgrad || a.dotx, "conv3x3 big: dot_actgrad without dotx");
    const bool g2 = a.x_hi_only != 0 && a.dotx != nullptr;       // the input-gradient instances only: a forward call keeps all three products
    OODGAN_REQUIRE(a.x_hi_only != 2 || g2, "conv3x3 big: hi-only input records (x_hi_only = 2) need dotx (the two-instruction instances)");
    if (g2) count_dispatch(OODGAN_DC_S1BIG_G2);
    if (g2 && a.x_hi_only == 2) count_dispatch(OODGAN_DC_S1BIG_XH);
    if (a.dot_actgrad && g2) hipLaunchKernelGGL((conv_f16s_s1big_kernel<true, true, false, true>), dim3((unsigned)total), dim3(512), BG_SMEM16, st, p, w16);
    else if (g2) hipLaunchKernelGGL((conv_f16s_s1big_kernel<true, false, false, true>), dim3((unsigned)total), dim3(512), BG_SMEM16, st, p, w16);
    else if (a.dot_actgrad) hipLaunchKernelGGL((conv_f16s_s1big_kernel<true, true>), dim3((unsigned)total), dim3(512), BG_SMEM16, st, p, w16);
    else if (a.dotx) hipLaunchKernelGGL((conv_f16s_s1big_kernel<true, false>), dim3((unsigned)total), dim3(512), BG_SMEM16, st, p, w16);
    else if (a.ys) hipLaunchKernelGGL((conv_f16s_s1big_kernel<false, false, true>), dim3((unsigned)total), dim3(512), BG_SMEM16, st, p, w16);
    else hipLaunchKernelGGL((conv_f16s_s1big_kernel<false, false>), dim3((unsigned)total), dim3(512), BG_SMEM16, st, p, w16);
    return check_launch("conv3x3_f16s_s1big");
}

}  // namespace oodgan

#ifdef OODGAN_CLOCK_STAMP
// stamp build only: buf = n pairs {shader cycles, 100 MHz ticks} of the K loop of workgroup blockIdx.x (last launch wins)
extern "C" int oodgan_debug_set_stamp_buffer(void* buf, long n) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_s1big_stamp), &p, sizeof(p)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_s1big_stamp_n), &n, sizeof(n)) != hipSuccess) {
        oodgan::set_error("debug_set_stamp_buffer: hipMemcpyToSymbol failed");
        return OODGAN_E_LAUNCH;
    }
    return 0;
}
#endif
