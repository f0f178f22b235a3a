// Split-f16 3x3 stride-1 conv for the >= 128-channel layers (64² ... 256² images): the matrix-bound middle of the
// generator (reference src/ops/StyleGAN/model.py:233-274, forward and input gradient).  Same arithmetic as
// conv_f16s_s1v2_kernel; different pipeline.  The v2 kernel keeps two 64 KB stages in LDS as two anti-phase groups and
// is bound by the LDS-DMA latency of one stage per step (each step moves 64 KB for 108 MFMAs per wave).  Here ONE
// workgroup of 8 waves owns a 16 x 32 pixel tile (twice the pixels per weight byte): a K stage is 39 KB of x + 36 KB
// of weights feeding 108 MFMAs in each of the 8 waves, two stages live in LDS, and the fetch of stage t+2 runs under
// the MFMAs of stage t+1; both waves of a SIMD issue MFMAs, so one covers the other's fragment-read latency.
//   * x tile 18 x 34 records of 64 B, slots rotated by (c>>2)&3 through the DMA source address (conflict-free reads,
//     no padding);  weights in the packed order [tap][hi|lo][k-half][64][8] (36 KB per 16 input channels);
//   * accumulators 2 M-tiles x 4 rows = 8 tiles per wave, MFMAs issued product-type-major so a tile is revisited
//     after 8 instructions;
//   * register epilogue: fp32 rows of 128 contiguous bytes per half wave, fused out-scale / noise / bias / lrelu, or
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
    int ablate;      // debug: 1 skip MFMAs, 2 skip the per-stage DMA
};

// PRE (input-gradient instance, oodgan_conv_args.dot_actgrad): the result is the gradient w.r.t. dotx, the OUTPUT of the
// up-sampling StyledConv below; the epilogue applies that layer's FusedLeakyReLU backward (fused_act.py:25-58) to the values
// it already holds — y <- dx * (dotx > 0 ? sqrt2 : 0.2 sqrt2) — so the blur^T producer that follows reads one tensor
// instead of two.  Three VALU operations per value: the kernel's waves have no spare issue slots for more (the sums of
// that layer's demodulation gradient are finished by the producer and the dot partials, include/oodgan.h).
template <bool DOT, int NW, bool PRE>
__global__ __launch_bounds__(64 * NW) void conv_f16s_s1big_kernel(
    const BigConv p, const uint4* __restrict__ wpk16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int mblk = w % p.mblocks;
    w /= p.mblocks;
    const int ntile = p.tiles_x * p.tiles_y;
    const int tile = w % ntile, b = w / ntile;
    const int ty = tile / p.tiles_x, tx = tile % p.tiles_x;
    const int r0 = ty * 16, c0 = tx * 32, m0 = mblk * 64;
    const int H = a.Hin, W = a.Win, M = a.M;

    // ---- per-lane DMA source offsets (bytes): x pieces relative to (plane of chunk 0, row r0, col c0), rotation applied;
    // weight pieces relative to the chunk's block.  Piece pc = wave + 4*i, i < 19 (pc < 75).
    constexpr int NT = 16 / NW;                              // rows per wave
    constexpr int NPW = (BG_PIECES + NW - 1) / NW;          // 19 (4 waves) / 10 (8 waves)
    unsigned off[NPW];
    const int KC = p.xd.KC;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int pc = wave + NW * i;
        if (pc < BG_XPIECES) {
            int P = pc * 64 + lane;
            if (P >= BG_XSLOTS) P = BG_XSLOTS - 1;
            const int row = P / (BG_C * 4), q = P % (BG_C * 4);
            const int c = q >> 2, s = ((q & 3) - ((c >> 2) & 3)) & 3;
            const int rr = min(r0 + row, p.xd.Hp - 1);      // tiles of 16 rows may reach below the 8-row padding
            off[i] = (unsigned)((((long)rr * p.xd.Wp + (c0 + c)) * 4 + s) * 16);
        } else {
            const int u = (pc - BG_XPIECES) * 64 + lane;    // 16-byte unit inside the 36 x 64 weight block
            const int row = u >> 6, j = u & 63;
            off[i] = (unsigned)((((long)row * p.Mp) + m0 + j) * 16);
        }
    }
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xs) + (long)b * KC * p.xd.plane * 16;
    const long xplane_bytes = p.xd.plane * 16;
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(wpk16);
    const long wchunk_bytes = (long)36 * p.Mp * 16;
    const int nchunk = (a.K + 15) / 16;

    // piece i of this wave for stage t (buffer buf); pieces are issued ONE AT A TIME between the MFMA groups of the
    // previous stage: a global_load_lds costs the issuing wave ~60-180 cycles (MI355X_MICROARCH.md), and a burst of ten of
    // them right after the barrier stalls every wave of the CU at once — DMA time and MFMA time then add up
    auto dma_piece = [&](int t, int buf, int i) {
        const int pc = wave + NW * i;
        if (pc >= BG_PIECES) return;
        unsigned char* dst = smem + buf * BG_STAGE;
        const unsigned char* src = (pc < BG_XPIECES ? xb + (long)t * xplane_bytes : wb + (long)t * wchunk_bytes) + off[i];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void*)(dst + pc * 1024), 16, 0, 0);
    };
    auto dma_stage = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) dma_piece(t, buf, i);
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // lane-constant fragment offsets
    unsigned lrd[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
            const int c = kx + l31;
            lrd[kx][lo] = c * 64 + (((half + 2 * lo + ((c >> 2) & 3)) & 3) << 4);
        }
    const unsigned lwf = (half * 64 + l31) * 16;

    // One stage = 9 taps on the same accumulators, taken column-major (kx outer): the three taps of a kernel column read the
    // SAME x rows (rows nt + ky of the column shift kx), so a column's NT + 2 rows are fetched from LDS once (hi and lo: 8
    // fragments for NT = 2) instead of NT per tap — 24 B-fragment reads per stage instead of 36, 60 ds_read_b128 instead of 72
    // with the 36 weight fragments.  The weight fragments of tap i+1 and, in halves, the x rows of the next column are
    // fetched while the MFMAs of tap i issue (two register sets each, order pinned with sched_barrier).  ONE barrier per
    // stage: after it every wave has finished stage t-1 (its buffer is free: the fetch of stage t+1 is issued right there and
    // runs under this stage's MFMAs) and stage t has landed (each wave waited for its own loads).
    struct AF { half8 ah[2], al[2]; };
    struct BF { half8 bh[NT + 2], bl[NT + 2]; };
    auto load_a = [&](AF& f, const unsigned char* lw, auto tp_c) {
        constexpr int tp = decltype(tp_c)::value;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f.ah[mt] = *reinterpret_cast<const half8*>(lw + (((tp * 2 + 0) * 2) * 64 + mt * 32) * 16);
            f.al[mt] = *reinterpret_cast<const half8*>(lw + (((tp * 2 + 1) * 2) * 64 + mt * 32) * 16);
        }
    };
    auto load_b = [&](BF& f, const unsigned char* lx, auto kx_c, auto r0_c, auto r1_c) {      // rows [r0, r1) of column kx
        constexpr int kx = decltype(kx_c)::value, ra = decltype(r0_c)::value, rb = decltype(r1_c)::value;
#pragma unroll
        for (int r = ra; r < rb; ++r) {
            f.bh[r] = *reinterpret_cast<const half8*>(lx + r * (BG_C * 64) + lrd[kx][0]);
            f.bl[r] = *reinterpret_cast<const half8*>(lx + r * (BG_C * 64) + lrd[kx][1]);
        }
    };
    auto mfma_tap = [&](const AF& fa, const BF& fb, auto ky_c) {
        constexpr int ky = decltype(ky_c)::value;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.ah[mt], fb.bh[nt + ky], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.ah[mt], fb.bl[nt + ky], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.al[mt], fb.bh[nt + ky], acc[mt][nt], 0, 0, 0);
    };
#define BG_IC(n) std::integral_constant<int, n>{}
#define BG_SB() __builtin_amdgcn_sched_barrier(0)
    static_assert(NT == 2, "the half-column prefetch below assumes four x rows per column");
    dma_stage(0, 0);
    for (int t = 0; t < nchunk; ++t) {
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): only stage t is outstanding
        __builtin_amdgcn_s_barrier();
        const bool pf = t + 1 < nchunk && !(p.ablate & 2);
        const int nb = (t + 1) & 1;
        if (p.ablate & 1) { if (pf) dma_stage(t + 1, nb); continue; }
        const unsigned char* lx = smem + (t & 1) * BG_STAGE + (wave * NT) * (BG_C * 64);
        const unsigned char* lw = smem + (t & 1) * BG_STAGE + BG_XBYTES + lwf;
        AF a0, a1;
        BF b0, b1;
        static_assert(NPW <= 10, "one DMA piece per tap slot below");
#define BG_DMA(i) if (pf && (i) < NPW) dma_piece(t + 1, nb, (i));
        // tap (ky, kx) has weight index ky*3 + kx
        load_b(b0, lx, BG_IC(0), BG_IC(0), BG_IC(4));
        load_a(a0, lw, BG_IC(0));
        BG_SB();
        load_a(a1, lw, BG_IC(3)); load_b(b1, lx, BG_IC(1), BG_IC(0), BG_IC(2)); BG_DMA(0) BG_SB(); mfma_tap(a0, b0, BG_IC(0)); BG_SB();
        load_a(a0, lw, BG_IC(6)); load_b(b1, lx, BG_IC(1), BG_IC(2), BG_IC(4)); BG_DMA(1) BG_SB(); mfma_tap(a1, b0, BG_IC(1)); BG_SB();
        load_a(a1, lw, BG_IC(1)); BG_DMA(2) BG_SB(); mfma_tap(a0, b0, BG_IC(2)); BG_SB();
        load_a(a0, lw, BG_IC(4)); load_b(b0, lx, BG_IC(2), BG_IC(0), BG_IC(2)); BG_DMA(3) BG_SB(); mfma_tap(a1, b1, BG_IC(0)); BG_SB();
        load_a(a1, lw, BG_IC(7)); load_b(b0, lx, BG_IC(2), BG_IC(2), BG_IC(4)); BG_DMA(4) BG_SB(); mfma_tap(a0, b1, BG_IC(1)); BG_SB();
        load_a(a0, lw, BG_IC(2)); BG_DMA(5) BG_SB(); mfma_tap(a1, b1, BG_IC(2)); BG_SB();
        load_a(a1, lw, BG_IC(5)); BG_DMA(6) BG_SB(); mfma_tap(a0, b0, BG_IC(0)); BG_SB();
        load_a(a0, lw, BG_IC(8)); BG_DMA(7) BG_SB(); mfma_tap(a1, b0, BG_IC(1)); BG_SB();
        BG_DMA(8) BG_DMA(9) BG_SB();
        mfma_tap(a0, b0, BG_IC(2));
#undef BG_DMA
    }
#undef BG_IC
#undef BG_SB

    // ---- epilogue from the accumulators
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
    const int px = c0 + l31;
    const float nw = (!DOT && a.noise) ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const float* nzb = a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * H * W : nullptr;
    float* red = reinterpret_cast<float*>(smem + BG_RED);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float osc[16], bia[16];
        float dsum[16];      // per 32-channel half: reduced right after its rows (registers)
#pragma unroll
        for (int r = 0; r < 16; ++r) dsum[r] = 0.f;
        unsigned moff[16], doff[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool mok = m < M;
            osc[r] = mok ? (a.out_scale ? a.out_scale[(long)b * a.out_scale_stride + m] : 1.f) * us : 0.f;
            bia[r] = (!DOT && a.bias && mok) ? a.bias[m] : 0.f;
            moff[r] = mok ? (unsigned)((long)m * p.out_plane * 4) : 0xFFFFFFFFu;
            doff[r] = mok ? (unsigned)((long)m * H * W * 4) : 0u;
        }
        const bool mfull = m0 + mt * 32 + 32 <= M;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int py = r0 + wave * NT + nt;
            const bool ok = py < H && px < W;
            float nz = 0.f;
            if (!DOT && nzb && ok) nz = nw * nzb[(long)py * W + px];
            unsigned char* yr = reinterpret_cast<unsigned char*>(a.y) + ((long)b * M * p.out_plane + (long)py * a.out_pitch + px) * 4;
            const unsigned char* dr = DOT ? reinterpret_cast<const unsigned char*>(a.dotx) + ((long)b * M * H * W + (long)py * W + px) * 4 : nullptr;
            float o[16], dv[16];
            if (DOT) {
                // all 16 loads of the row are issued before the first use (one conditional load per value would serialise
                // them into 16 memory latencies)
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
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[mt][nt][r];
                o[r] = v * osc[r];
                if (DOT) {
                    dsum[r] += (v * us) * dv[r];
                    if (PRE) o[r] *= dv[r] > 0.f ? kSqrt2 : 0.2f * kSqrt2;
                } else {
                    o[r] += nz + bia[r];
                    if (a.act == OODGAN_ACT_LRELU) o[r] = (o[r] > 0.f ? o[r] : 0.2f * o[r]) * kSqrt2;
                }
            }
            if (ok) {
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
        if (DOT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = half_sum_dpp(dsum[r]);
                if (l31 == kHalfSumLane) red[wave * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = v;
            }
        }
    }
    if (DOT) {
        // the host sizes the partial-sum tables for 8-row tiles: this 16-row tile owns two of their slots
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
    if (!(a.mode == OODGAN_CONV_S1 && a.x_sform && a.K >= min_k && a.M >= 64 && a.ys == nullptr && a.y != nullptr &&
          (a.act == OODGAN_ACT_NONE || a.act == OODGAN_ACT_LRELU) && a.in_scale == nullptr && a.in_shift == nullptr &&
          !(a.dotx && (a.noise || a.bias || a.act != OODGAN_ACT_NONE))))
        return false;
    // enough 16x32 tiles to fill the chip; smaller layers keep the latency-oriented instances
    const long items = (long)((a.Hin + 15) / 16) * ((a.Win + 31) / 32) * a.B * ((a.M + 63) / 64);
    const char* e = getenv("OODGAN_S1_BIG_MIN_ITEMS");      // tests lower the threshold to reach this kernel with small tensors
    return items >= (e ? atol(e) : 256);
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
    if (a.dotx) {
        OODGAN_REQUIRE(a.dot_part != nullptr, "conv3x3: dotx without dot_part");
        OODGAN_REQUIRE(a.dot_nparts == ((a.Hin + 7) / 8) * p.tiles_x, "conv3x3 f16s S1: dot_nparts %d != %d", a.dot_nparts,
                       ((a.Hin + 7) / 8) * p.tiles_x);
    }
    OODGAN_REQUIRE((long)a.M * p.out_plane * 4 < (1L << 32) && (long)a.M * a.Hin * a.Win * 4 < (1L << 32), "conv3x3 big: plane too large");
#ifdef OODGAN_DEBUG_ABLATE      // profiling builds only: the ablation bits make the kernel skip work (wrong results)
    static const int abl = getenv("OODGAN_BIG_ABLATE") ? atoi(getenv("OODGAN_BIG_ABLATE")) : 0;
    p.ablate = abl;
#else
    p.ablate = 0;
#endif
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<true, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<true, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_s1big_kernel<false, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BG_SMEM), true);
    (void)once;
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
    // 8 waves: two per SIMD keep the MFMA pipe fed while the partner waits on LDS (the 4-wave variant was slower than the
    // v2 tile kernel: 474 / 475 / 552 us against 393 / 407 / 464 on the 64² / 128² / 256² layers)
    if (a.dot_actgrad) {
        OODGAN_REQUIRE(a.dotx, "conv3x3 big: dot_actgrad without dotx");
        hipLaunchKernelGGL((conv_f16s_s1big_kernel<true, 8, true>), dim3((unsigned)total), dim3(512), BG_SMEM, st, p, w16);
    } else if (a.dotx) hipLaunchKernelGGL((conv_f16s_s1big_kernel<true, 8, false>), dim3((unsigned)total), dim3(512), BG_SMEM, st, p, w16);
    else hipLaunchKernelGGL((conv_f16s_s1big_kernel<false, 8, false>), dim3((unsigned)total), dim3(512), BG_SMEM, st, p, w16);
    return check_launch("conv3x3_f16s_s1big");
}

}  // namespace oodgan
