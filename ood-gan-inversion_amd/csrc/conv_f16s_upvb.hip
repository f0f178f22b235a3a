// The up-sampling StyledConv of the highest level in ONE pass (round 4): conv_transpose2d(stride 2, pad 0) + Blur(pad = (1,1)) + noise +
// bias + leaky ReLU (reference src/ops/StyleGAN/model.py:199-205,247-258,343-350; Blur = upfirdn2d with the flipped 4x4 kernel,
// src/ops/op/upfirdn2d.py:160-193), split-f16 arithmetic, S-form input, F-form output.
//
// Until round 3 this layer (64 -> 32 channels, 512² -> 1024², batch 8) was two passes: conv_f16s_t2v2_kernel<1> wrote the
// (2H+1) x (2W+1) transposed-conv result z (1.08 GB) and blur_act_fform_strip read it back: 0.44 + 0.46 ms per W+ step, both bound by
// memory.  z never needs to exist.  The blur kernel is rank one, kf[a][b] = kv[a] * kh[b], so
//
//   y[Y][X] = sum_b kh[b] * v[Y][X + b - 1],      v[Y][X'] = sum_a kv[a] * z[Y + a - 1][X']          (vertical pass first)
//   v[2I + py][2j + kx] = sum_{d = -1,0,1} x[I + d][j] * Wv[py][d][kx],
//   Wv[py][d][kx] = sum_{a, ky : py + a - 1 - ky = 2d} kv[a] * W[ky][kx]                             (built once per layer by the host)
//
// i.e. the VERTICAL blur is folded into the weights — two 3x3 weight sets, one per output row parity, each a stride-1 conv along the
// rows and the transposed stride-2 conv along the columns: 18 taps per 16-channel chunk instead of 9, twice the matrix work of the
// transposed conv on a layer whose matrix work is small (0.15 TFLOP algorithmic) — and the HORIZONTAL 4-tap pass runs on the
// accumulators: a lane holds the two column phases of position j, the neighbouring positions are the neighbouring lanes (DPP).  No
// vertical halo, no z, no LDS exchange.  Folding both directions into the weights would cost four times the matrix work.
//
// Workgroup = NW waves = NW position rows x 32 positions x 32 output channels (wave = row, v_mfma_f32_32x32x16_f16, accumulators = the 4
// output phases of a position); the x tile is NW + 2 rows x 33 positions (one more row above and below than the transposed conv needs).
// The first version was the 4-wave tile kernel of conv_f16s_v2.hip with the second weight set (697 us at batch 8): every 4-row tile fetched
// 36 KB of weights per 16-channel chunk next to 13 KB of x — 3.8 GB of LDS-DMA per launch against 1.6 GB of tensor traffic.  Two forms
// are built (tunable "upvb_waves"):
//   12: one PERSISTENT 12-wave workgroup per CU walking its tiles with two 73 KB stages in LDS, one barrier per stage (stage t+1 — or
//       the first stage of the NEXT tile — is requested before stage t is computed); 2.1x fewer fetched bytes, but the epilogue
//       (vector work + 197 KB of stores per tile) and the K loop of a CU never overlap: 624 us in the loop
//       (ablation, standalone: DMA alone 170 us, matrix instructions 340, epilogue 326, all 762);
//    6: two 6-wave workgroups per CU, one 57 KB stage each, one tile per workgroup: the vector epilogue of one runs beside the matrix
//       instructions of the other (separate pipes), at 1.6x the fetched bytes of the 12-wave form.
// Positions c0 .. c0+31 give the 60 output columns 2(c0+1) .. 2(c0+30)+1: the horizontal pass needs the neighbours of a position, so
// consecutive tiles overlap by two positions (c0 = 30 tx - 1).
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>

using namespace oodgan;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int UV_REC = 80;                                   // LDS bytes per position: 4 slots + 1 pad slot (conv_f16s_v2.hip)
constexpr int UV_C = 33;
constexpr int UV_MB = 32;
constexpr int UV_WROWS = 36;                                 // rows of one packed 3x3 set per chunk: [tap][hi|lo][k-half]
constexpr int UV_WSET = UV_WROWS * UV_MB * 16;               // 18432 bytes per set and chunk in LDS
constexpr int UV_WPIECES = 2 * UV_WSET / 1024;               // 36
constexpr int UV_STRIDE = 30;                                // positions a tile advances
constexpr int UV_CST_MAX = 3072;                             // floats of the persistent form's constant table: B * M * 2 + M must fit (12 KB)
#ifndef UPVB_ABL
#define UPVB_ABL 0           // profiling builds (-DUPVB_ABL=bits, wrong results): 1 skip the matrix instructions, 2 skip the per-stage DMA, 4 skip the epilogue
#endif

template <int NW>
struct UVCfg {
    static constexpr int R = NW + 2, NPOS = R * UV_C;
    static constexpr int XPIECES = (NPOS * 5 + 63) / 64;     // one-KiB DMA pieces of the x tile (NW = 12: 37, 6: 21)
    static constexpr int XBYTES = XPIECES * 1024;
    static constexpr int PIECES = XPIECES + UV_WPIECES;      // 73 / 57
    static constexpr int STAGE = PIECES * 1024;              // 74752 / 58368
    static constexpr int NPW = (PIECES + NW - 1) / NW;       // pieces per wave and stage (7 / 10)
    static constexpr int SMEM_PERSIST = 2 * STAGE + UV_CST_MAX * 4;
    static constexpr int SMEM_TILE = STAGE + 3 * UV_MB * 4;
};

struct UpVB {
    const uint4* xs;
    SDims xd;
    const uint4* wpk;        // two packed sets, the second `wset_units` 16-byte units behind the first
    long wset_units;
    const float* unscale4;   // {2^-e0, 2^e0, 2^-e1, 2^e1}
    const float* kh;         // the four horizontal taps as the kernel applies them (flipped), device
    const float* out_scale; int out_scale_stride;
    const float* bias;
    const float* noise; int noise_batch; const float* noise_w;
    const float* ys_scale; int ys_scale_stride;
    unsigned* vmax;
    float* y;                // F-form (B, M/16, 2H, 2W, 16)
    int B, K, M, H, W, act;
    int tiles_x, tiles_y, mblocks, Mp;
    int total;               // work items: tiles x images x 32-channel blocks
};

template <int CTRL>
__device__ __forceinline__ float wave_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// per-lane DMA source offsets (bytes) of this wave's pieces for tile (r0, c0, m0).  x: tile origin = image (r0-1, c0-1) = padded
// (r0, c0); indices clamped into the padded plane (row 0 / column 0 and the last row / column are zeros: what lies outside the image
// contributes nothing).  The lane's place inside the tile (row, column, slot of each x piece; row and channel of each weight piece) does
// not depend on the tile: `upvb_lane_pieces` computes it ONCE per workgroup (integer divisions by 5 and 33 per piece: ~60 vector
// instructions each — recomputed per tile they were a quarter of the kernel's vector instructions, profiles/r4_upvb_pmc.csv);
// `upvb_offsets` then costs ~8 per piece.
template <int NW>
struct UVLane { int a[UVCfg<NW>::NPW]; };                    // x piece: r << 16 | c << 4 | s; weight piece: 16-byte units of (set, row, j)

template <int NW>
__device__ __forceinline__ void upvb_lane_pieces(const UpVB& p, int wave, int lane, UVLane<NW>& L) {
    using C = UVCfg<NW>;
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
        const int pc = wave + NW * i;
        if (pc < C::XPIECES) {
            int P = pc * 64 + lane;
            if (P >= C::NPOS * 5) P = C::NPOS * 5 - 1;
            const int pos = P / 5;
            int s = P % 5;
            if (s == 4) s = 0;
            L.a[i] = (pos / UV_C) << 16 | (pos % UV_C) << 4 | s;
        } else {
            const int pw = pc - C::XPIECES;                  // 0..35: set = pw / 18
            const int set = pw / 18, u = (pw % 18) * 64 + lane;
            L.a[i] = (int)((long)set * p.wset_units + (long)(u / UV_MB) * p.Mp + u % UV_MB);
        }
    }
}

template <int NW>
__device__ __forceinline__ void upvb_offsets(const UpVB& p, int wave, const UVLane<NW>& L, int r0, int c0, int m0, unsigned (&off)[UVCfg<NW>::NPW]) {
    using C = UVCfg<NW>;
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
        const int pc = wave + NW * i;
        if (pc < C::XPIECES) {
            const int r = L.a[i] >> 16, c = (L.a[i] >> 4) & 0xFFF, s = L.a[i] & 15;
            const int rr = min(max(r0 + r, 0), p.xd.Hp - 1), cc = min(max(c0 + c, 0), p.xd.Wp - 1);
            off[i] = (unsigned)(((rr * p.xd.Wp + cc) * 4 + s) * 16);
        } else {
            off[i] = (unsigned)((L.a[i] + m0) * 16);
        }
    }
}

template <int NW>
__device__ __forceinline__ void upvb_dma(const UpVB& p, int wave, const unsigned char* xb, const unsigned (&off)[UVCfg<NW>::NPW], int t,
                                         unsigned char* dst) {
    using C = UVCfg<NW>;
    const long wchunk_bytes = (long)UV_WROWS * p.Mp * 16;
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
        const int pc = wave + NW * i;
        if (pc < C::PIECES) {
            const unsigned char* src = (pc < C::XPIECES ? xb + (long)t * p.xd.plane * 16 : reinterpret_cast<const unsigned char*>(p.wpk) + (long)t * wchunk_bytes) + off[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void*)(dst + pc * 1024), 16, 0, 0);
        }
    }
}

// one piece of a stage (persistent form: the pieces of stage t+1 are issued one at a time BETWEEN the tap steps of stage t — a burst of
// seven global_load_lds right after the barrier keeps every wave of the CU in the issue of its loads at the same moment)
template <int NW>
__device__ __forceinline__ void upvb_dma_piece(const UpVB& p, int wave, const unsigned char* xb, const unsigned (&off)[UVCfg<NW>::NPW], int t,
                                               unsigned char* dst, int i) {
    using C = UVCfg<NW>;
    const int pc = wave + NW * i;
    if (pc < C::PIECES) {
        const long wchunk_bytes = (long)UV_WROWS * p.Mp * 16;
        const unsigned char* src = (pc < C::XPIECES ? xb + (long)t * p.xd.plane * 16 : reinterpret_cast<const unsigned char*>(p.wpk) + (long)t * wchunk_bytes) + off[i];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void*)(dst + pc * 1024), 16, 0, 0);
    }
}

// 18 taps of one 16-channel chunk, row by row of the x tile: row tap d = ky' - 1 (x rows I-1, I, I+1 = tile rows wave .. wave+2), weight set py,
// column tap kx -> phase py*2 + (kx & 1), x column j - (kx >> 1) (tile columns l31 + 1 - (kx >> 1)).  The fragments of step i+1 are read
// under the matrix instructions of step i (two register sets, order pinned); one x row (4 fragments) is live at a time.
template <int NW, typename Between>
__device__ __forceinline__ void upvb_stage(const unsigned char* lx, int wave, int l31, int half, f32x16 (&acc)[4], Between between) {
    using C = UVCfg<NW>;
    const unsigned char* lwh = lx + C::XBYTES + (half * UV_MB + l31) * 16;
    const unsigned char* lxh = lx + (wave * UV_C + l31) * UV_REC + half * 16;
#define XFRAG(posoff, lo_) (*reinterpret_cast<const half8*>(lxh + (posoff) * UV_REC + (lo_) * 32))
#define WFRAG(set, tap, lo_) (*reinterpret_cast<const half8*>(lwh + (set) * UV_WSET + ((((tap) * 2 + (lo_)) * 2) * UV_MB) * 16))
#define MFMA3(accv, ah, al, bh, bl)                                              \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accv, 0, 0, 0);        \
    accv = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accv, 0, 0, 0);
    half8 ah[2], al[2], bh[2][2], bl[2][2];
    ah[0] = WFRAG(0, 0, 0);
    al[0] = WFRAG(0, 0, 1);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        bh[0][cc] = XFRAG(cc, 0);
        bl[0][cc] = XFRAG(cc, 1);
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const int kyp = i / 6, py = (i / 3) & 1, kx = i % 3;        // step i = (kyp, py, kx)
        const int cur = i & 1, rb = kyp & 1;
        if (i + 1 < 18) {
            const int kyn = (i + 1) / 6, pyn = ((i + 1) / 3) & 1, kxn = (i + 1) % 3;
            ah[cur ^ 1] = WFRAG(pyn, kyn * 3 + kxn, 0);
            al[cur ^ 1] = WFRAG(pyn, kyn * 3 + kxn, 1);
        }
        if (i % 6 == 3 && kyp < 2) {                         // the next x row, half a row of taps ahead
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                bh[rb ^ 1][cc] = XFRAG((kyp + 1) * UV_C + cc, 0);
                bl[rb ^ 1][cc] = XFRAG((kyp + 1) * UV_C + cc, 1);
            }
        }
        between(i);
        __builtin_amdgcn_sched_barrier(0);
        MFMA3(acc[py * 2 + (kx & 1)], ah[cur], al[cur], bh[rb][1 - (kx >> 1)], bl[rb][1 - (kx >> 1)]);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef XFRAG
#undef WFRAG
#undef MFMA3
}

// Epilogue on the accumulators: lane (l31, half) holds, for position j = c0 + l31 of row I, channels (r & 3) + 8 (r >> 2) + 4 half.
// cst_o / cst_b / cst_y: LDS tables (out scale, bias, ys scale) of the tile's 32 channels.  Returns true if the wave issued its stores.
// Vector work per value pair, kept small because the waves of a CU run it beside (or between) their matrix phases: sqrt2 of the
// activation folded into the scale and the noise / bias term (lrelu * sqrt2 = max(u, 0.2 u) with u = sqrt2 * v), one v_max3 per pair for
// the range maximum (scaled by |ys| once per channel at the end).
__device__ __forceinline__ bool upvb_epilogue(const UpVB& p, f32x16 (&acc)[4], int b, int m0, int I, int J, int l31, int half,
                                              const float* cst_o, const float* cst_b, const float* cst_y, float& vm) {
    const int H = p.H, Ho = 2 * H, Wo = 2 * p.W;
    const long HWo = (long)Ho * Wo;
    const float kh0 = p.kh[0], kh1 = p.kh[1], kh2 = p.kh[2], kh3 = p.kh[3];
    const bool lrelu = p.act == OODGAN_ACT_LRELU;
    const float g = lrelu ? kSqrt2 : 1.f;
    const float us0 = p.unscale4[0] * g, us1 = p.unscale4[2] * g;
    const float nw = (p.noise ? (p.noise_w ? p.noise_w[0] : 1.f) : 0.f) * g;
    const bool lane_ok = l31 >= 1 && l31 <= UV_STRIDE && I < H && 2 * J + 1 < Wo && J >= 0;
    // noise of the four output pixels (2I+py, 2J+px): one float2 per row from a clamped address
    const float* np = p.noise ? p.noise + (long)(p.noise_batch > 1 ? b : 0) * HWo : p.kh;
    float2 nz[2];
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        const long o = (long)min(2 * I + py, Ho - 1) * Wo + min(max(2 * J, 0), Wo - 2);
        nz[py] = *reinterpret_cast<const float2*>(np + (p.noise ? o : 0));
        nz[py].x *= nw;
        nz[py].y *= nw;
    }
    const int KC = (p.M + 15) / 16;
    float* yb = p.y + (long)b * KC * HWo * 16;
    float tvm = 0.f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {                        // registers 4 g4 .. 4 g4 + 3: four consecutive channels = one 16-byte piece
        float o[2][2][4];                                    // [py][px][channel of the quad]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * g4 + e;
            const int cl = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float sc = cst_o[cl], bv = cst_b[cl] * g, ys = fabsf(cst_y[cl]);
            float mr = 0.f;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const float s = sc * (py ? us1 : us0);
                const float v0 = acc[py * 2 + 0][r] * s, v1 = acc[py * 2 + 1][r] * s;       // v columns 2J, 2J + 1 (x sqrt2 with the activation)
                const float l1 = wave_dpp<0x138>(v1);        // wave_shr:1 — column 2J - 1 from position J - 1
                const float r0_ = wave_dpp<0x130>(v0);       // wave_shl:1 — columns 2J + 2, 2J + 3 from position J + 1
                const float r1_ = wave_dpp<0x130>(v1);
                float y0 = kh0 * l1 + kh1 * v0 + kh2 * v1 + kh3 * r0_ + (nz[py].x + bv);
                float y1 = kh0 * v0 + kh1 * v1 + kh2 * r0_ + kh3 * r1_ + (nz[py].y + bv);
                if (lrelu) {
                    y0 = fmaxf(y0, 0.2f * y0);
                    y1 = fmaxf(y1, 0.2f * y1);
                }
                o[py][0][e] = y0;
                o[py][1][e] = y1;
                asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mr) : "v"(y0), "v"(y1));       // one instruction per pair for the range maximum
            }
            tvm = fmaxf(tvm, mr * ys);                       // (M %% 32 == 0: every channel of the block exists; lanes masked once, below)
        }
        const int mq = m0 + 8 * g4 + 4 * half;               // first channel of the quad
        if (lane_ok && mq < p.M) {
            const int kc = mq >> 4, qd = (mq & 15) >> 2;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    float* dst = yb + (((long)kc * Ho + 2 * I + py) * Wo + 2 * J + px) * 16 + 4 * qd;
                    if (UPVB_ABL & 8)        // profiling: the same bytes as 1 KB contiguous per instruction (wrong layout)
                        dst = yb + (((long)kc * Ho + 2 * I + py) * Wo + 2 * (J - l31)) * 16 + ((g4 & 1) * 2 + px) * 256 + (half * 32 + l31) * 4;
                    if (UPVB_ABL & 16) {     // profiling: no stores (the values stay live through vm)
                        vm += o[py][px][0] + o[py][px][1] + o[py][px][2] + o[py][px][3];
                        continue;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(o[py][px][0], o[py][px][1], o[py][px][2], o[py][px][3]);
                }
        }
    }
    vm = fmaxf(vm, lane_ok ? tvm : 0.f);
    return I < H;
}

// ---- persistent form: one NW-wave workgroup per CU, two stages in LDS
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_f16s_upvb_kernel(const UpVB p) {
    using C = UVCfg<NW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int ntile = p.tiles_x * p.tiles_y;
    // every XCD (blockIdx & 7) takes a contiguous chunk of the work list; inside the chunk the XCD's workgroups stride through it together
    int it, it_end, it_step;
    {
        const int xcd = blockIdx.x & 7, q = p.total >> 3, r = p.total & 7;
        const int first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        it = first + (int)(blockIdx.x >> 3);
        it_end = first + q + (xcd < r ? 1 : 0);
        it_step = (int)(gridDim.x >> 3);
    }
    if (it >= it_end) return;
    const int nchunk = (p.K + 15) / 16;
    unsigned off[C::NPW];                                    // DMA state of the tile whose stages are being REQUESTED
    const unsigned char* xb_dma;
    UVLane<NW> LP;
    upvb_lane_pieces<NW>(p, wave, lane, LP);
    auto set_dma_tile = [&](int item) {
        int w = item;
        const int mblk_ = w % p.mblocks;
        w /= p.mblocks;
        const int tile_ = w % ntile, b_ = w / ntile;
        upvb_offsets<NW>(p, wave, LP, (tile_ / p.tiles_x) * NW, (tile_ % p.tiles_x) * UV_STRIDE - 1, mblk_ * UV_MB, off);
        xb_dma = reinterpret_cast<const unsigned char*>(p.xs) + (long)b_ * p.xd.KC * p.xd.plane * 16;
    };
    float* ctab = reinterpret_cast<float*>(smem + 2 * C::STAGE);
    float vm = 0.f;
    int vb = -1;
    set_dma_tile(it);
    upvb_dma<NW>(p, wave, xb_dma, off, 0, smem);
    // per-channel constants of every sample -> LDS, once per workgroup (a global load per tile would sit behind the epilogue's stores in
    // the wave's memory queue): [B][M] out scale, [B][M] ys scale, [M] bias; read behind the K loop's barriers
    {
        const int BM = p.B * p.M;
        for (int i = tid; i < 2 * BM + p.M; i += 64 * NW) {
            float v;
            if (i < BM) v = p.out_scale ? p.out_scale[(long)(i / p.M) * p.out_scale_stride + i % p.M] : 1.f;
            else if (i < 2 * BM) v = p.ys_scale ? p.ys_scale[(long)((i - BM) / p.M) * p.ys_scale_stride + (i - BM) % p.M] : 1.f;
            else v = p.bias ? p.bias[i - 2 * BM] : 0.f;
            ctab[i] = v;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): written before this wave arrives at the first barrier
    }
    bool prev_full = false;                                  // the previous tile's epilogue issued its full set of memory operations
    int gs = 0;                                              // stage counter of the whole tile stream: LDS buffer = gs & 1
    for (;;) {
        const int itn = it + it_step;
        const bool has_next = itn < it_end;
        int mblk, tile, b;
        {
            int w = it;
            mblk = w % p.mblocks;
            w /= p.mblocks;
            tile = w % ntile;
            b = w / ntile;
        }
        const int r0 = (tile / p.tiles_x) * NW, c0 = (tile % p.tiles_x) * UV_STRIDE - 1, m0 = mblk * UV_MB;
        f32x16 acc[4];                                       // output phase py * 2 + px
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int t = 0; t < nchunk; ++t, ++gs) {
            // Single barrier per stage: after it every wave has finished stage gs-1 (its buffer is free) and stage gs has landed (each
            // wave waited for its own pieces).  At a tile's first stage the wave's vector-memory queue holds, oldest first, those pieces
            // (requested under the previous tile's last stage) and then that tile's epilogue: 2 noise loads and 16 stores; vmcnt counts
            // in issue order, so "all but the 18 youngest" covers the pieces without waiting for the stores to drain.
            if (t == 0 && prev_full) __builtin_amdgcn_s_waitcnt(0x4F72);      // vmcnt(18)
            else __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            const bool last = t + 1 == nchunk;
            if (last && has_next) set_dma_tile(itn);         // from here on the DMA state describes the NEXT tile
            const bool pf = (!last || has_next) && !(UPVB_ABL & 2);
            const int tn = last ? 0 : t + 1;
            unsigned char* nbuf = smem + ((gs + 1) & 1) * C::STAGE;
            if (UPVB_ABL & 1) {
                if (pf) upvb_dma<NW>(p, wave, xb_dma, off, tn, nbuf);
                continue;
            }
            upvb_stage<NW>(smem + (gs & 1) * C::STAGE, wave, l31, half, acc, [&](int i) {
                if (pf && i < C::NPW) upvb_dma_piece<NW>(p, wave, xb_dma, off, tn, nbuf, i);
            });
        }
        if (!(UPVB_ABL & 4)) {
            // the range maximum is recorded per sample: flush when the walk moves on to another image
            if (p.vmax && vb >= 0 && vb != b) {
                record_vmax(p.vmax, vb, vm);
                vm = 0.f;
            }
            vb = b;
            prev_full = upvb_epilogue(p, acc, b, m0, r0 + wave, c0 + l31, l31, half, ctab + b * p.M + m0, ctab + 2 * p.B * p.M + m0,
                                      ctab + p.B * p.M + b * p.M + m0, vm);
        }
        if (!has_next) break;
        it = itn;
    }
    if (p.vmax && vb >= 0) record_vmax(p.vmax, vb, vm);
}

// ---- tile form: one tile per NW-wave workgroup, ONE stage in LDS, several workgroups per CU cover each other's DMA waits and epilogues
template <int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_f16s_upvb_tile_kernel(const UpVB p) {
    using C = UVCfg<NW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int ntile = p.tiles_x * p.tiles_y;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int mblk = w % p.mblocks;
    w /= p.mblocks;
    const int tile = w % ntile, b = w / ntile;
    const int r0 = (tile / p.tiles_x) * NW, c0 = (tile % p.tiles_x) * UV_STRIDE - 1, m0 = mblk * UV_MB;
    const int nchunk = (p.K + 15) / 16;
    unsigned off[C::NPW];
    UVLane<NW> LP;
    upvb_lane_pieces<NW>(p, wave, lane, LP);
    upvb_offsets<NW>(p, wave, LP, r0, c0, m0, off);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xs) + (long)b * p.xd.KC * p.xd.plane * 16;
    float* ctab = reinterpret_cast<float*>(smem + C::STAGE);      // [3][32]: out scale, bias, ys scale of this tile's channels
    float cv[3] = {1.f, 0.f, 1.f};
    if (tid < UV_MB) {
        const int m = min(m0 + tid, p.M - 1);
        if (p.out_scale) cv[0] = p.out_scale[(long)b * p.out_scale_stride + m];
        if (p.bias) cv[1] = p.bias[m];
        if (p.ys_scale) cv[2] = p.ys_scale[(long)b * p.ys_scale_stride + m];
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int t = 0; t < nchunk; ++t) {
        if (t > 0) __builtin_amdgcn_s_barrier();             // every wave is done with the previous stage
        if (!(UPVB_ABL & 2) || t == 0) upvb_dma<NW>(p, wave, xb, off, t, smem);
        if (t == 0 && tid < UV_MB) {
            ctab[tid] = cv[0];
            ctab[UV_MB + tid] = cv[1];
            ctab[2 * UV_MB + tid] = cv[2];
        }
        __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (UPVB_ABL & 1) continue;
        upvb_stage<NW>(smem, wave, l31, half, acc, [](int) {});
    }
    if (UPVB_ABL & 4) return;
    float vm = 0.f;
    upvb_epilogue(p, acc, b, m0, r0 + wave, c0 + l31, l31, half, ctab, ctab + UV_MB, ctab + 2 * UV_MB, vm);
    if (p.vmax) record_vmax(p.vmax, b, vm);
}

}  // namespace

// CU count + the kernels' dynamic-LDS attributes, once per process (one device per process: common.hpp, bound_device_ok)
static int upvb_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_upvb_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, UVCfg<12>::SMEM_PERSIST) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_upvb_tile_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, UVCfg<6>::SMEM_TILE) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_upvb_tile_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, UVCfg<4>::SMEM_TILE) != hipSuccess)
            return 0;
        return cus;
    }();
    return n;
}

// 1 when oodgan_upconv_vblur_fform takes this shape
extern "C" int oodgan_upconv_vblur_supported(int B, int K, int M, int H, int W) {
    if (!(B > 0 && K >= 16 && (K % 16) == 0 && M >= 32 && (M % 32) == 0 && H >= 4 && W >= 30 && 2L * B * M + M <= UV_CST_MAX)) return 0;
    const SDims xd = sform_dims(K, H, W);
    if (xd.plane * 16 >= (1L << 31)) return 0;
    return oodgan::bound_device_ok("upconv_vblur_supported") && upvb_num_cus() > 0 ? 1 : 0;
}

extern "C" int oodgan_upconv_vblur_fform(const void* xs, const void* wpk2, long wset_bytes, const float* unscale4, const float* kh4,
                                         const float* out_scale, int out_scale_stride, const float* bias, const float* noise,
                                         int noise_batch, const float* noise_w, int act, const float* ys_scale, int ys_scale_stride,
                                         unsigned* vmax, float* y, int B, int K, int M, int H, int W, void* stream) {
    OODGAN_REQUIRE(xs && wpk2 && unscale4 && kh4 && y, "upconv_vblur_fform: null tensor");
    OODGAN_REQUIRE(oodgan_upconv_vblur_supported(B, K, M, H, W), "upconv_vblur_fform: unsupported shape (K %% 16, M %% 32, H >= 4, W >= 30)");
    OODGAN_REQUIRE(act == OODGAN_ACT_NONE || act == OODGAN_ACT_LRELU, "upconv_vblur_fform: act must be none or lrelu");
    OODGAN_REQUIRE(noise == nullptr || noise_batch == 1 || noise_batch == B, "upconv_vblur_fform: noise_batch");
    OODGAN_REQUIRE((wset_bytes & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0, "upconv_vblur_fform: alignment");
    const int nw = (int)oodgan::tunable(oodgan::OODGAN_TUN_UPVB_WAVES);
    OODGAN_REQUIRE(nw == 12 || nw == 6 || nw == 4, "upconv_vblur_fform: tunable upvb_waves must be 12, 6 or 4");
    UpVB p;
    p.xs = reinterpret_cast<const uint4*>(xs);
    p.xd = sform_dims(K, H, W);
    p.wpk = reinterpret_cast<const uint4*>(wpk2);
    p.wset_units = wset_bytes / 16;
    p.unscale4 = unscale4; p.kh = kh4;
    p.out_scale = out_scale; p.out_scale_stride = out_scale_stride;
    p.bias = bias; p.noise = noise; p.noise_batch = noise_batch; p.noise_w = noise_w;
    p.ys_scale = ys_scale; p.ys_scale_stride = ys_scale_stride; p.vmax = vmax; p.y = y;
    p.B = B; p.K = K; p.M = M; p.H = H; p.W = W; p.act = act;
    p.tiles_x = (2 * W + 2 * UV_STRIDE - 1) / (2 * UV_STRIDE);
    p.tiles_y = (H + nw - 1) / nw;
    p.mblocks = M / UV_MB;
    p.Mp = (M + 63) / 64 * 64;
    const long total = (long)p.tiles_x * p.tiles_y * B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "upconv_vblur_fform: too many tiles");
    p.total = (int)total;
    // set up on the first oodgan_upconv_vblur_supported() call (the engine makes one when it prepares the weights): nothing here
    // happens for the first time inside a stream capture
    OODGAN_REQUIRE(upvb_num_cus() > 0, "upconv_vblur_fform: no device");
    oodgan::count_dispatch(oodgan::OODGAN_DC_UPVB);
    hipStream_t st = oodgan::as_stream(stream);
    if (nw == 12) {
        // persistent grid: one workgroup per CU, a multiple of 8 so that every XCD owns a contiguous chunk of tiles
        long per_xcd = (total + 7) / 8;
        if (per_xcd > upvb_num_cus() / 8) per_xcd = upvb_num_cus() / 8 > 0 ? upvb_num_cus() / 8 : 1;
        hipLaunchKernelGGL(conv_f16s_upvb_kernel<12>, dim3((unsigned)(8 * per_xcd)), dim3(64 * 12), UVCfg<12>::SMEM_PERSIST, st, p);
    } else if (nw == 6) {
        hipLaunchKernelGGL(conv_f16s_upvb_tile_kernel<6>, dim3((unsigned)total), dim3(64 * 6), UVCfg<6>::SMEM_TILE, st, p);
    } else {
        hipLaunchKernelGGL(conv_f16s_upvb_tile_kernel<4>, dim3((unsigned)total), dim3(64 * 4), UVCfg<4>::SMEM_TILE, st, p);
    }
    return oodgan::check_launch("upconv_vblur_fform");
}
