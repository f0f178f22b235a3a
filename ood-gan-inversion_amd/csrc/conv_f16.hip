// fp16 modulated 3x3 conv for the high-resolution, low-channel layers (BASELINE.json configs[4], SURVEY.md §8 C5:
// x (16,32,1024,1024) f16, 32 -> 32 channels) — the HBM-bound end of ModulatedConv2d.forward
// (reference src/ops/StyleGAN/model.py:233-245,268-274) with the StyledConv tail (noise, bias, leaky-ReLU*sqrt2,
// model.py:283-292,343-350) fused in.  f16 operands, one MFMA per product, fp32 accumulate, f16 result.
//
// Layout ("H-form"):  X[b][kc = ceil(C/16)][Hp][Wp][16 x f16]   32-byte record per pixel and 16-channel block,
// pixel (y,x) at [y+1][x+1], zero border and padding exactly as the S-form (sform.hpp) so a halo'd tile row is one
// contiguous 1088-byte run.  Output is written in the same form (the next layer's input).
//
// Like the reference, the modulation/demodulation is folded into PER-SAMPLE weights
//   w[b,co,ci,tap] = demod[b,co] * scale * W[co,ci,tap] * s[b,ci]          (model.py:236-241)
// (B x 18 KB at 32x32 channels — negligible), packed in MFMA A-fragment order by modconv_f16_pack_kernel.
//
// Kernel = persistent 256-thread workgroups, a few per CU.  A workgroup keeps its sample's whole weight tensor in
// REGISTERS (18 A fragments), walks a contiguous range of 8x32-pixel tiles of its XCD and double-buffers the input
// tile (10x34 records x 2 channel blocks = 21.8 KB) in LDS by LDS-DMA: the fetch of tile i+1 is in flight while
// tile i is multiplied (36 MFMA 32x32x16 per wave) and written.  The epilogue never touches LDS: the accumulator
// layout gives each lane 8 of a pixel's 16 channels, one v_permlane32_swap pair completes the 16-byte half record,
// and a wave stores 1 KB contiguous per 16-channel block.
#include "common.hpp"
#include "sform.hpp"
#include <cstdint>
#include <cstdlib>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

struct HDims {
    int KC, Hp, Wp;
    long plane;          // 16-byte units per (b,kc) plane
};

__host__ __device__ inline HDims hform_dims(int C, int H, int W) {
    HDims d;
    d.KC = (C + 15) / 16;
    d.Hp = (H + 1 + 7) / 8 * 8 + 2;
    d.Wp = (W + 1 + 31) / 32 * 32 + 2;
    d.plane = (long)d.Hp * d.Wp * 2;
    return d;
}

constexpr int IN_R = 10, IN_C = 34, NPOS = IN_R * IN_C;      // halo'd 8x32 tile
constexpr int SLOTS_PER_KC = NPOS * 2;                       // 16-byte slots per channel block

template <int KC>
struct Geo {
    static constexpr int slots = KC * SLOTS_PER_KC;
    static constexpr int pieces = ((slots + 63) / 64 + 3) / 4 * 4;   // 1 KiB DMA pieces, the same number for every wave
    static constexpr int buf_bytes = pieces * 1024;                  // (KC=2: 24; the tail pieces land in padding)
    static constexpr int ppw = pieces / 4;                           // pieces per wave
    static constexpr int noise_off = pieces * 1024;                  // + one KiB: the tile's 8x32 fp32 noise values
    static constexpr int buf_total = noise_off + 1024;
};

struct MCArgs {
    const uint4* x;          // H-form input
    const uint4* wpk;        // packed per-sample weights
    const float* noise;      // (noise_batch, H, W) fp32 or null
    const float* noise_w;    // device scalar or null (= 1)
    const float* bias;       // (M) or null
    uint4* y;                // H-form output
    int noise_batch, act;
    int B, K, M, H, W;
    int tiles_x, tiles_y;
    int flags;               // bit0: do not wait for the previous tile's stores at the top of the loop
    HDims xd, yd;
};

template <int KC, bool LRELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void modconv_f16_kernel(const MCArgs p) {
    using G = Geo<KC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // ---- this workgroup's contiguous tile range inside its XCD's chunk
    const int ntile = p.tiles_x * p.tiles_y;
    const int T = ntile * p.B;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int LB = ((int)gridDim.x - xcd + 7) >> 3;
    const int q = T >> 3, rem = T & 7;
    const int start = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    const int cnt = q + (xcd < rem ? 1 : 0);
    if (lb >= cnt) return;

    // ---- constants of the epilogue.  With the leaky ReLU the sqrt(2) gain is already in the packed weights
    // (oodgan_modconv_f16_pack) and is folded into bias and noise weight here: sqrt2*lrelu(v) = lrelu(sqrt2*v), and
    // lrelu(u) = max(u, 0.2u), so the activation costs two VALU operations per value.
    const float gain = LRELU ? kSqrt2 : 1.f;
    const float* bias_p = p.bias ? p.bias : reinterpret_cast<const float*>(p.wpk);
    float bias_g[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // an unconditional load from a clamped channel (a valid address when there is no bias), masked by a select: `m < M ? bias[m]
        // : 0` is a branch, a load and a vmcnt(0) per value — sixteen round trips in a row before the first tile is requested
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float bv = bias_p[min(m, p.M - 1)];
        bias_g[r] = (p.bias && m < p.M) ? gain * bv : 0.f;
    }
    const float nwg = p.noise ? gain * (p.noise_w ? p.noise_w[0] : 1.f) : 0.f;
    const int MC = (p.M + 15) / 16;
    const float* nzp = p.noise ? p.noise : reinterpret_cast<const float*>(p.x);
    const bool use_vm4 = p.flags & 1;

    // ---- per-lane DMA source offsets relative to the tile origin (16-byte units)
    long xoff[G::ppw];
#pragma unroll
    for (int i = 0; i < G::ppw; ++i) {
        int P = (wave + 4 * i) * 64 + lane;
        if (P >= G::slots) P = G::slots - 1;
        const int kc = P / SLOTS_PER_KC, qq = P % SLOTS_PER_KC;
        const int pos = qq >> 1, s = qq & 1;
        const int r = pos / IN_C, c = pos % IN_C;
        xoff[i] = (long)kc * p.xd.plane + ((long)r * p.xd.Wp + c) * 2 + s;
    }
    auto tile_coords = [&](int item, int& b, int& r0, int& c0) {
        const int t = start + item;
        b = t / ntile;
        const int tt = t - b * ntile;
        r0 = (tt / p.tiles_x) * 8;
        c0 = (tt % p.tiles_x) * 32;
    };
    auto dma = [&](int item, int buf) {
        int b, r0, c0;
        tile_coords(item, b, r0, c0);
        const uint4* base = p.x + (long)b * KC * p.xd.plane + ((long)r0 * p.xd.Wp + c0) * 2;
        unsigned char* dst = smem + buf * G::buf_total;
#pragma unroll
        for (int i = 0; i < G::ppw; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xoff[i]),
                                             (lds_void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
        {   // noise: each wave fetches the two rows it will use itself, one dword per lane, clamped inside the image
            const int ny = min(r0 + wave * 2 + half, p.H - 1), nx = min(c0 + l31, p.W - 1);
            const float* src = nzp + ((long)(p.noise_batch > 1 ? b : 0) * p.H + ny) * p.W + nx;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (lds_void*)(dst + G::noise_off + wave * 256), 4, 0, 0);
        }
    };

    half8 areg[9][KC];
    int cur_b = -1;

    // s_waitcnt immediates (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8); issued through the
    // builtin so that the compiler's own wait-count bookkeeping sees them
    constexpr int kVm0 = 0x0F70, kVm4 = 0x0F74;
    __builtin_amdgcn_s_waitcnt(kVm0);        // bias / noise-weight loads: retire them here, not inside the loop
    dma(lb, 0);
    int it = 0;
    bool prev_full = false;
    for (int item = lb; item < cnt; item += LB, ++it) {
        // the prefetched tile must have landed (LDS-DMA is not covered by the barrier).  Memory operations retire
        // in issue order, so when the previous tile issued exactly its 4 stores AFTER the prefetch, vmcnt(4) already
        // guarantees the DMA; the stores themselves may still be in flight.
        if (prev_full) __builtin_amdgcn_s_waitcnt(kVm4);
        else __builtin_amdgcn_s_waitcnt(kVm0);
        __builtin_amdgcn_s_barrier();        // plain barrier: __syncthreads() would add its own vmcnt(0) fence
        int b, r0, c0;
        tile_coords(item, b, r0, c0);
        const int px = c0 + l31;
        // always issued (the last iteration re-fetches its own tile into the idle buffer)
        if (!(p.flags & 4)) dma(item + LB < cnt ? item + LB : item, (it + 1) & 1);
        prev_full = use_vm4 && MC == 2 && r0 + 8 <= p.H && c0 + 32 <= p.W;
        if (b != cur_b) {
            cur_b = b;
            const half8* wb = reinterpret_cast<const half8*>(p.wpk) + (long)b * 9 * KC * 64;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) areg[tp][kc] = wb[((tp * KC + kc) * 2 + half) * 32 + l31];
            __builtin_amdgcn_s_waitcnt(kVm0);      // once per sample; keeps the wait out of the common path
        }
        const unsigned char* lbuf = smem + (it & 1) * G::buf_total;
        const unsigned char* lx = lbuf + ((wave * 2) * IN_C + l31) * 32 + half * 16;
        // accumulators start at bias + noise (the tile's noise values arrived with the tile)
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float nz = reinterpret_cast<const float*>(lbuf + G::noise_off)[(wave * 2 + nt) * 32 + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = fmaf(nwg, nz, bias_g[r]);
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            half8 bf[4][3];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    bf[rr][kx] = *reinterpret_cast<const half8*>(lx + kc * (SLOTS_PER_KC * 16) + (rr * IN_C + kx) * 32);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[tp][kc], bf[nt + tp / 3][tp % 3], acc[nt], 0, 0, 0);
        }
        // ---- epilogue straight from the accumulators
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int py = r0 + wave * 2 + nt;
            const bool ok = py < p.H && px < p.W;
            unsigned pk[8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                float o[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v = acc[nt][2 * r2 + e];
                    o[e] = LRELU ? fmaxf(v, 0.2f * v) : v;
                }
                if (p.M < 32) {          // partial channel block: keep the padding channels at zero
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 2 * r2 + e;
                        if ((r & 3) + 8 * (r >> 2) + 4 * half >= p.M) o[e] = 0.f;
                    }
                }
                half2v h;                 // round-to-nearest-even pair conversion (v_cvt_pk_f16_f32 on gfx950)
                h[0] = (_Float16)o[0];
                h[1] = (_Float16)o[1];
                pk[r2] = __builtin_bit_cast(unsigned, h);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (cb >= MC) break;
                // lanes 0-31 hold channels {0-3, 8-11} of the block, lanes 32-63 {4-7, 12-15}: complete the halves
                auto s0 = __builtin_amdgcn_permlane32_swap(pk[cb * 4 + 0], pk[cb * 4 + 2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(pk[cb * 4 + 1], pk[cb * 4 + 3], false, false);
                if (ok && !(p.flags & 2)) {
                    uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    p.y[((long)b * MC + cb) * p.yd.plane + ((long)(py + 1) * p.yd.Wp + (px + 1)) * 2 + half] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Strip kernel (32 input channels): one workgroup walks DOWN a 32-pixel-wide strip of the image, so the two halo
// rows a tile shares with its upper neighbour are already in LDS — every input row is fetched once per strip.
//   LDS ring of 32 image rows x [kc 2][34 records][32 B] (2176 B per row) = 4 groups of 8 rows; group g holds the
//   rows r' in [8g, 8g+8), r' = (padded row) - (first padded row of the segment) - 2, so tile t needs the last two
//   rows of group t-1 and all of group t.  Groups are fetched two ahead (3 x 17 KB in flight per workgroup).
//   Inside a row the two 16-byte halves of record c are stored swapped when (c>>3)&1: with 32-byte records the
//   ds_read_b128 lane groups would otherwise hit every bank twice.  The swap is done for free by the per-lane
//   SOURCE address of the LDS-DMA.
constexpr int RG_ROW_BYTES = 2 * IN_C * 32;          // 2176
constexpr int RG_GROUP_SLOTS = 8 * RG_ROW_BYTES / 16;  // 1088 = 17 pieces
constexpr int RG_PIECES = 17;
constexpr int RG_RING_BYTES = 32 * RG_ROW_BYTES;      // 69632
constexpr int RG_NOISE_OFF = RG_RING_BYTES;           // 4 x 1 KiB noise tiles
constexpr int RG_SMEM = RG_RING_BYTES + 4096;

struct StripArgs {
    MCArgs m;
    int seg_tiles;        // tiles per work item (a strip is cut into ceil(tiles_y/seg_tiles) segments)
    int nseg;
};

template <bool LRELU>
__global__ __launch_bounds__(256) void modconv_f16_strip_kernel(const StripArgs sa) {
    const MCArgs& p = sa.m;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // ---- work item: (b, segment, strip column); neighbouring columns go to the same XCD and run together
    int w = blockIdx.x;
    {
        const int total = gridDim.x, xcd = w & 7, idx = w >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = w % p.tiles_x;
    const int seg = (w / p.tiles_x) % sa.nseg;
    const int b = w / (p.tiles_x * sa.nseg);
    const int t0 = seg * sa.seg_tiles;
    const int n = min(sa.seg_tiles, p.tiles_y - t0);
    const int c0 = tx * 32;
    const int R0 = 8 * t0;                                  // first image row = first padded row of the segment

    // ---- epilogue constants (see modconv_f16_kernel)
    const float gain = LRELU ? kSqrt2 : 1.f;
    const float* bias_p = p.bias ? p.bias : reinterpret_cast<const float*>(p.wpk);
    float bias_g[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // unconditional loads, masked by a select (see modconv_f16_kernel): one round trip together with the weights
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float bv = bias_p[min(m, p.M - 1)];
        bias_g[r] = (p.bias && m < p.M) ? gain * bv : 0.f;
    }
    const float nwg = p.noise ? gain * (p.noise_w ? p.noise_w[0] : 1.f) : 0.f;
    const int MC = (p.M + 15) / 16;
    const float* nzb = (p.noise ? p.noise : reinterpret_cast<const float*>(p.x)) + (long)(p.noise_batch > 1 ? b : 0) * p.H * p.W;

    // ---- weights: the whole per-sample tensor in registers
    half8 areg[9][2];
    {
        const half8* wb = reinterpret_cast<const half8*>(p.wpk) + (long)b * 9 * 2 * 64;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) areg[tp][kc] = wb[((tp * 2 + kc) * 2 + half) * 32 + l31];
    }

    // ---- per-lane DMA source offsets inside a group (bytes relative to the group's first row at column c0)
    // piece pc = wave + 4*i (i < 5; only wave 0 has a fifth piece)
    unsigned xoff[5];
    int xrow[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int P = (wave + 4 * i) * 64 + lane;
        if (P >= RG_GROUP_SLOTS) P = RG_GROUP_SLOTS - 1;
        const int row = P / 136, q = P % 136;
        const int kc = q / 68, c2 = q % 68;
        const int c = c2 >> 1, s = (c2 & 1) ^ ((c >> 3) & 1);
        xrow[i] = row;
        xoff[i] = (unsigned)(((long)kc * p.xd.plane + ((long)row * p.xd.Wp + c) * 2 + s) * 16);
    }
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + ((long)b * 2 * p.xd.plane + (long)c0 * 2) * 16;
    const long row_bytes = (long)p.xd.Wp * 32;
    const int npc = wave == 0 ? 5 : 4;

    auto dma_group = [&](int g) {            // g in [0, n): rows R0 + 2 + 8g .. +7 (padded), ring group g & 3
        const int gg = min(g, n - 1);        // past the end: re-fetch the last group into a dead ring group
        const unsigned char* base = xb + (long)(R0 + 2 + 8 * gg) * row_bytes;
        unsigned char* dst = smem + (g & 3) * (8 * RG_ROW_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xoff[i]),
                                             (lds_void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xoff[4]),
                                             (lds_void*)(dst + 16 * 1024), 16, 0, 0);
        // noise of tile g: each wave its own two rows, one dword per lane, clamped inside the image
        const int ny = min(R0 + 8 * gg + wave * 2 + half, p.H - 1), nx = min(c0 + l31, p.W - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(nzb + (long)ny * p.W + nx),
                                         (lds_void*)(smem + RG_NOISE_OFF + (g & 3) * 1024 + wave * 256), 4, 0, 0);
    };

    constexpr int kVm0 = 0x0F70;
    __builtin_amdgcn_s_waitcnt(kVm0);        // weights, bias: retired here, never inside the loop
    {   // prologue: the two halo rows above the first tile (ring rows 30,31 = tail of group "-1"), then groups 0, 1
        const int Rm = R0 + 2 - 8;           // first padded row of group -1 (may be negative: clamp per lane)
        unsigned char* dst = smem + 3 * (8 * RG_ROW_BYTES);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i == 4 && wave != 0) break;
            const int rr = max(Rm + xrow[i], 0) - (Rm + xrow[i]);       // rows above the image -> row 0 (never used)
            const unsigned char* src = xb + (long)Rm * row_bytes + xoff[i] + (long)rr * row_bytes;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (lds_void*)(dst + (wave + 4 * i) * 1024), 16, 0, 0);
        }
        dma_group(0);
        dma_group(1);
    }

    // lane-constant parts of the LDS read addresses (with the half swap) and of the store offsets
    unsigned lrd[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) lrd[kx] = (kx + l31) * 32 + ((half ^ (((kx + l31) >> 3) & 1)) << 4);
    unsigned lst[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) lst[nt] = ((unsigned)((wave * 2 + nt + 1) * p.yd.Wp + l31 + 1) * 2 + half) * 16;
    unsigned char* yb = reinterpret_cast<unsigned char*>(p.y) + ((long)b * MC * p.yd.plane + ((long)R0 * p.yd.Wp + c0) * 2) * 16;
    const long yrow8 = (long)p.yd.Wp * 32 * 8, ycb = p.yd.plane * 16;
    const bool col_full = c0 + 32 <= p.W && MC == 2 && (p.flags & 1);
    const int px = c0 + l31;
    int ragged = 3;                           // bit0/bit1: one of the last two tiles did not issue its 4 stores

    for (int t = 0; t < n; ++t) {
        // group t (issued two iterations ago) must have landed.  Memory operations retire in issue order; behind it
        // there may be: the stores of tile t-2 (4), group t+1 with its noise (npc+1), the stores of tile t-1 (4).
        if (t == 0) {
            if (wave == 0) __builtin_amdgcn_s_waitcnt(0x0F76); else __builtin_amdgcn_s_waitcnt(0x0F75);
        } else if (ragged) {
            __builtin_amdgcn_s_waitcnt(kVm0);
        } else {
            if (wave == 0) __builtin_amdgcn_s_waitcnt(0x0F7E); else __builtin_amdgcn_s_waitcnt(0x0F7D);
        }
        __builtin_amdgcn_s_barrier();
        if (!(p.flags & 4)) dma_group(t + 2);
        const bool full = col_full && R0 + 8 * t + 8 <= p.H;
        ragged = ((ragged << 1) | (full ? 0 : 1)) & 3;
        if (t == 0) ragged |= 2;             // tile -1 issued no stores: iteration 1 cannot use the counted wait

        const unsigned char* nzl = smem + RG_NOISE_OFF + (t & 3) * 1024;
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float nz = reinterpret_cast<const float*>(nzl)[(wave * 2 + nt) * 32 + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = fmaf(nwg, nz, bias_g[r]);
        }
        unsigned rbase[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) rbase[rr] = ((8 * t + 30 + 2 * wave + rr) & 31) * RG_ROW_BYTES;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            half8 bf[4][3];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    bf[rr][kx] = *reinterpret_cast<const half8*>(smem + rbase[rr] + kc * (IN_C * 32) + lrd[kx]);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[tp][kc], bf[nt + tp / 3][tp % 3], acc[nt], 0, 0, 0);
        }
        unsigned char* yt = yb + (long)t * yrow8;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int py = R0 + 8 * t + wave * 2 + nt;
            const bool ok = py < p.H && px < p.W;
            unsigned pk[8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                float o[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v = acc[nt][2 * r2 + e];
                    o[e] = LRELU ? fmaxf(v, 0.2f * v) : v;
                }
                if (p.M < 32) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 2 * r2 + e;
                        if ((r & 3) + 8 * (r >> 2) + 4 * half >= p.M) o[e] = 0.f;
                    }
                }
                half2v h;
                h[0] = (_Float16)o[0];
                h[1] = (_Float16)o[1];
                pk[r2] = __builtin_bit_cast(unsigned, h);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                if (cb >= MC) break;
                auto s0 = __builtin_amdgcn_permlane32_swap(pk[cb * 4 + 0], pk[cb * 4 + 2], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(pk[cb * 4 + 1], pk[cb * 4 + 3], false, false);
                if (ok && !(p.flags & 2))
                    *reinterpret_cast<uint4*>(yt + cb * ycb + lst[nt]) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
    }
}

// per-sample packed weights: out[b][tap][kc][half][m 32][8 f16], value demod[b,m]*scale*W[m,k,tap]*s[b,k].
// latent != NULL: the style is computed here — the modulation EqualLinear of ModulatedConv2d (model.py:223,236: s = latent @
// (mod_w / sqrt(S))^T + mod_b) — so that the op is two launches (pack, conv) instead of three.  Eight threads per style
// entry / per output channel, sums over the fixed lane order of an 8-lane butterfly (deterministic).
__global__ __launch_bounds__(256) void modconv_f16_pack_kernel(const float* __restrict__ w, const float* __restrict__ style,
                                                               int style_stride, float scale, int demodulate, float gain,
                                                               half8* __restrict__ out, int B, int M, int K, int KC,
                                                               const float* __restrict__ latent, int latent_stride,
                                                               const float* __restrict__ mod_w, const float* __restrict__ mod_b, int S) {
    __shared__ float dm[32], sl[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int row = tid >> 3, sub = tid & 7;           // 32 rows x 8 threads
    if (latent) {
        float acc = 0.f;
        if (row < K) {
            const float* lp = latent + (long)b * latent_stride;
            const float* wp = mod_w + (long)row * S;
            for (int j = sub; j < S; j += 8) acc += lp[j] * wp[j];
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (sub == 0) sl[row] = row < K ? acc * rsqrtf((float)S) + (mod_b ? mod_b[row] : 0.f) : 0.f;
    } else if (tid < 32) {
        sl[tid] = tid < K ? style[(long)b * style_stride + tid] : 0.f;
    }
    __syncthreads();
    {
        float acc = 0.f;
        if (demodulate && row < M) {
            for (int k = sub; k < K; k += 8) {
                float wsq = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) { const float v = w[((long)row * K + k) * 9 + t]; wsq += v * v; }
                acc += sl[k] * sl[k] * wsq;
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (sub == 0) dm[row] = (demodulate && row < M) ? rsqrtf(scale * scale * acc + 1e-8f) : 1.f;
    }
    __syncthreads();
    const int n = 9 * KC * 2 * 32;
    for (int u = tid; u < n; u += 256) {
        const int m = u & 31, hf = (u >> 5) & 1, kc = (u >> 6) % KC, tp = (u >> 6) / KC;
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kc * 16 + hf * 8 + j;
            float f = 0.f;
            if (m < M && k < K) f = gain * dm[m] * scale * w[((long)m * K + k) * 9 + tp] * sl[k];
            v[j] = (_Float16)f;
        }
        out[(long)b * n + u] = v;
    }
}

__global__ __launch_bounds__(256) void to_hform_kernel(const float* __restrict__ x, uint4* __restrict__ out, int B, int C, int H,
                                                       int W, HDims d) {
    const long total = (long)B * d.KC * H * W;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int xx = (int)(e % W), yy = (int)((e / W) % H);
        const int kc = (int)((e / ((long)W * H)) % d.KC), b = (int)(e / ((long)W * H * d.KC));
        half8 h[2];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int c = kc * 16 + j;
            h[j >> 3][j & 7] = (_Float16)(c < C ? x[(((long)b * C + c) * H + yy) * W + xx] : 0.f);
        }
        half8* o = reinterpret_cast<half8*>(out + ((long)b * d.KC + kc) * d.plane + ((long)(yy + 1) * d.Wp + xx + 1) * 2);
        o[0] = h[0];
        o[1] = h[1];
    }
}

__global__ __launch_bounds__(256) void from_hform_kernel(const uint4* __restrict__ in, float* __restrict__ y, int B, int C, int H,
                                                         int W, HDims d) {
    const long total = (long)B * C * H * W;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int xx = (int)(e % W), yy = (int)((e / W) % H);
        const int c = (int)((e / ((long)W * H)) % C), b = (int)(e / ((long)W * H * C));
        const _Float16* rec = reinterpret_cast<const _Float16*>(in + ((long)b * d.KC + (c >> 4)) * d.plane +
                                                                ((long)(yy + 1) * d.Wp + xx + 1) * 2);
        y[e] = (float)rec[c & 15];
    }
}

}  // namespace

extern "C" long oodgan_hform_bytes(int B, int C, int H, int W) {
    const HDims d = hform_dims(C, H, W);
    return (long)B * d.KC * d.plane * 16;
}

extern "C" int oodgan_to_hform(const float* x, void* out, int B, int C, int H, int W, void* stream) {
    OODGAN_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "to_hform: bad args");
    const HDims d = hform_dims(C, H, W);
    const long total = (long)B * d.KC * H * W;
    hipLaunchKernelGGL(to_hform_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), x,
                       reinterpret_cast<uint4*>(out), B, C, H, W, d);
    return check_launch("to_hform");
}

extern "C" int oodgan_from_hform(const void* in, float* y, int B, int C, int H, int W, void* stream) {
    OODGAN_REQUIRE(in && y && B > 0 && C > 0 && H > 0 && W > 0, "from_hform: bad args");
    const HDims d = hform_dims(C, H, W);
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(from_hform_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const uint4*>(in), y, B, C, H, W, d);
    return check_launch("from_hform");
}

extern "C" long oodgan_modconv_f16_wbytes(int B, int M, int K) {
    (void)M;
    return (long)B * 9 * ((K + 15) / 16) * 2 * 32 * 16;
}

extern "C" int oodgan_modconv_f16_pack(const float* weight, const float* style, int style_stride, float scale, int demodulate,
                                       int act, void* wpk, int B, int M, int K, void* stream) {
    OODGAN_REQUIRE(weight && style && wpk && B > 0, "modconv_f16_pack: bad args");
    OODGAN_REQUIRE(M >= 1 && M <= 32 && K >= 1 && K <= 32, "modconv_f16: supports up to 32 -> 32 channels (got %d -> %d)", K, M);
    hipLaunchKernelGGL(modconv_f16_pack_kernel, dim3(B), dim3(256), 0, as_stream(stream), weight, style, style_stride, scale,
                       demodulate, act == OODGAN_ACT_LRELU ? kSqrt2 : 1.f, reinterpret_cast<half8*>(wpk), B, M, K, (K + 15) / 16,
                       (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, 0);
    return check_launch("modconv_f16_pack");
}

extern "C" int oodgan_modconv_f16_pack_affine(const float* weight, const float* latent, int latent_stride, const float* mod_weight,
                                              const float* mod_bias, int S, float scale, int demodulate, int act, void* wpk, int B,
                                              int M, int K, void* stream) {
    OODGAN_REQUIRE(weight && latent && mod_weight && wpk && B > 0 && S > 0, "modconv_f16_pack_affine: bad args");
    OODGAN_REQUIRE(M >= 1 && M <= 32 && K >= 1 && K <= 32, "modconv_f16: supports up to 32 -> 32 channels (got %d -> %d)", K, M);
    hipLaunchKernelGGL(modconv_f16_pack_kernel, dim3(B), dim3(256), 0, as_stream(stream), weight, (const float*)nullptr, 0, scale,
                       demodulate, act == OODGAN_ACT_LRELU ? kSqrt2 : 1.f, reinterpret_cast<half8*>(wpk), B, M, K, (K + 15) / 16,
                       latent, latent_stride, mod_weight, mod_bias, S);
    return check_launch("modconv_f16_pack_affine");
}

extern "C" int oodgan_modconv_f16(const void* x, const void* wpk, const float* noise, int noise_batch, const float* noise_w,
                                  const float* bias, int act, void* y, int B, int K, int M, int H, int W, void* stream) {
    OODGAN_REQUIRE(x && wpk && y && B > 0 && H > 0 && W > 0, "modconv_f16: bad args");
    OODGAN_REQUIRE(M >= 1 && M <= 32 && K >= 1 && K <= 32, "modconv_f16: supports up to 32 -> 32 channels (got %d -> %d)", K, M);
    OODGAN_REQUIRE(act == OODGAN_ACT_NONE || act == OODGAN_ACT_LRELU, "modconv_f16: act must be none or lrelu");
    if (!oodgan::bound_device_ok("modconv_f16")) return OODGAN_E_ARG;
    MCArgs p;
    p.x = reinterpret_cast<const uint4*>(x);
    p.wpk = reinterpret_cast<const uint4*>(wpk);
    p.noise = noise; p.noise_batch = noise_batch; p.noise_w = noise_w; p.bias = bias; p.act = act;
    p.y = reinterpret_cast<uint4*>(y);
    p.B = B; p.K = K; p.M = M; p.H = H; p.W = W;
    p.tiles_y = (H + 7) / 8;
    p.tiles_x = (W + 31) / 32;
    p.xd = hform_dims(K, H, W);
    p.yd = hform_dims(M, H, W);
#ifdef OODGAN_DEBUG_ABLATE      // profiling builds only: 2 no stores, 4 no loads (wrong results)
    static const int ablate = getenv("OODGAN_F16_ABLATE") ? atoi(getenv("OODGAN_F16_ABLATE")) : 0;
#else
    const int ablate = 0;
#endif
    p.flags = 1 | ablate;          // bit 0: counted vmcnt waits
    const long T = (long)p.tiles_x * p.tiles_y * B;
    OODGAN_REQUIRE(T < (1L << 31), "modconv_f16: too many tiles");
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            set_error("modconv_f16: cannot query the device");
            return OODGAN_E_LAUNCH;
        }
        num_cu = prop.multiProcessorCount;
    }
    const int KC = (K + 15) / 16;
#define OODGAN_LAUNCH(KC_, LR_)                                                                                            \
    {                                                                                                                   \
        constexpr int sm = 2 * Geo<KC_>::buf_total;                                                                     \
        static int occ = 0;                                                                                             \
        if (!occ) {                                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_f16_kernel<KC_, LR_>),                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, sm);                                  \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, modconv_f16_kernel<KC_, LR_>, 256, sm) != hipSuccess || occ < 1) \
                occ = 1;                                                                                                \
        }                                                                                                               \
        long g = (long)num_cu * occ;                                                                                    \
        g = (g + 7) / 8 * 8;                                                                                            \
        if (g > T) g = T;                                                                                               \
        hipLaunchKernelGGL((modconv_f16_kernel<KC_, LR_>), dim3((unsigned)g), dim3(256), sm, as_stream(stream), p);          \
    }
    const bool lr = act == OODGAN_ACT_LRELU;
    if (KC == 2) {
        // strips of 32 columns; cut into segments only when there are fewer strips than ~2 workgroups per CU
        StripArgs sa;
        sa.m = p;
        const long strips = (long)B * p.tiles_x;
        int nseg = (int)((2L * num_cu + strips - 1) / strips);
        if (nseg < 1) nseg = 1;
        int seg_tiles = (p.tiles_y + nseg - 1) / nseg;
        if (seg_tiles < 4) seg_tiles = p.tiles_y < 4 ? p.tiles_y : 4;
        sa.seg_tiles = seg_tiles;
        sa.nseg = (p.tiles_y + seg_tiles - 1) / seg_tiles;
        const long nblk = strips * sa.nseg;
        OODGAN_REQUIRE(nblk < (1L << 31), "modconv_f16: grid too large");
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_f16_strip_kernel<true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, RG_SMEM),
                            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_f16_strip_kernel<false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, RG_SMEM), true);
        (void)once;
        if (lr) hipLaunchKernelGGL((modconv_f16_strip_kernel<true>), dim3((unsigned)nblk), dim3(256), RG_SMEM, as_stream(stream), sa);
        else hipLaunchKernelGGL((modconv_f16_strip_kernel<false>), dim3((unsigned)nblk), dim3(256), RG_SMEM, as_stream(stream), sa);
        return check_launch("modconv_f16_strip");
    }
    if (lr) OODGAN_LAUNCH(1, true) else OODGAN_LAUNCH(1, false)            // <= 16 input channels: the tile-order persistent kernel
#undef OODGAN_LAUNCH
    return check_launch("modconv_f16");
}
