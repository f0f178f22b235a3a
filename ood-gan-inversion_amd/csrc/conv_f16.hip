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
    static constexpr int pieces = (slots + 63) / 64;         // 1 KiB DMA pieces (KC=2: 22)
    static constexpr int buf_bytes = pieces * 1024;
    static constexpr int ppw = (pieces + 3) / 4;             // pieces per wave
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
    HDims xd, yd;
};

template <int KC>
__global__ __launch_bounds__(256) void modconv_f16_kernel(const MCArgs p) {
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
        unsigned char* dst = smem + buf * G::buf_bytes;
#pragma unroll
        for (int i = 0; i < G::ppw; ++i) {
            const int pc = wave + 4 * i;
            if (pc < G::pieces)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + xoff[i]),
                                                 (lds_void*)(dst + pc * 1024), 16, 0, 0);
        }
    };

    // ---- constants of the epilogue
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        bias_r[r] = (p.bias && m < p.M) ? p.bias[m] : 0.f;
    }
    const float nw = p.noise ? (p.noise_w ? p.noise_w[0] : 1.f) : 0.f;
    const int MC = (p.M + 15) / 16;

    half8 areg[9][KC];
    int cur_b = -1;

    dma(lb, 0);
    int it = 0;
    for (int item = lb; item < cnt; item += LB, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // LDS-DMA is not covered by the barrier
        __syncthreads();
        if (item + LB < cnt) dma(item + LB, (it + 1) & 1);
        int b, r0, c0;
        tile_coords(item, b, r0, c0);
        if (b != cur_b) {
            cur_b = b;
            const half8* wb = reinterpret_cast<const half8*>(p.wpk) + (long)b * 9 * KC * 64;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) areg[tp][kc] = wb[((tp * KC + kc) * 2 + half) * 32 + l31];
        }
        const unsigned char* lx = smem + (it & 1) * G::buf_bytes + ((wave * 2) * IN_C + l31) * 32 + half * 16;
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
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
        const int px = c0 + l31;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int py = r0 + wave * 2 + nt;
            const bool ok = py < p.H && px < p.W;
            float nz = 0.f;
            if (p.noise && ok) nz = nw * p.noise[((long)(p.noise_batch > 1 ? b : 0) * p.H + py) * p.W + px];
            unsigned pk[8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                float o[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * r2 + e;
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[nt][r] + nz + bias_r[r];
                    if (p.act == OODGAN_ACT_LRELU) v = (v > 0.f ? v : 0.2f * v) * kSqrt2;
                    o[e] = m < p.M ? v : 0.f;
                }
                half2v h;
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
                if (ok) {
                    uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    p.y[((long)b * MC + cb) * p.yd.plane + ((long)(py + 1) * p.yd.Wp + (px + 1)) * 2 + half] = v;
                }
            }
        }
    }
}

// per-sample packed weights: out[b][tap][kc][half][m 32][8 f16], value demod[b,m]*scale*W[m,k,tap]*s[b,k]
__global__ __launch_bounds__(256) void modconv_f16_pack_kernel(const float* __restrict__ w, const float* __restrict__ style,
                                                               int style_stride, float scale, int demodulate,
                                                               half8* __restrict__ out, int B, int M, int K, int KC) {
    __shared__ float dm[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = style + (long)b * style_stride;
    if (tid < 32) {
        float d = 1.f;
        if (demodulate && tid < M) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) {
                float wsq = 0.f;
                for (int t = 0; t < 9; ++t) { const float v = w[((long)tid * K + k) * 9 + t]; wsq += v * v; }
                acc += s[k] * s[k] * wsq;
            }
            d = rsqrtf(scale * scale * acc + 1e-8f);
        }
        dm[tid] = d;
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
            if (m < M && k < K) f = dm[m] * scale * w[((long)m * K + k) * 9 + tp] * s[k];
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
                                       void* wpk, int B, int M, int K, void* stream) {
    OODGAN_REQUIRE(weight && style && wpk && B > 0, "modconv_f16_pack: bad args");
    OODGAN_REQUIRE(M >= 1 && M <= 32 && K >= 1 && K <= 32, "modconv_f16: supports up to 32 -> 32 channels (got %d -> %d)", K, M);
    hipLaunchKernelGGL(modconv_f16_pack_kernel, dim3(B), dim3(256), 0, as_stream(stream), weight, style, style_stride, scale,
                       demodulate, reinterpret_cast<half8*>(wpk), B, M, K, (K + 15) / 16);
    return check_launch("modconv_f16_pack");
}

extern "C" int oodgan_modconv_f16(const void* x, const void* wpk, const float* noise, int noise_batch, const float* noise_w,
                                  const float* bias, int act, void* y, int B, int K, int M, int H, int W, void* stream) {
    OODGAN_REQUIRE(x && wpk && y && B > 0 && H > 0 && W > 0, "modconv_f16: bad args");
    OODGAN_REQUIRE(M >= 1 && M <= 32 && K >= 1 && K <= 32, "modconv_f16: supports up to 32 -> 32 channels (got %d -> %d)", K, M);
    OODGAN_REQUIRE(act == OODGAN_ACT_NONE || act == OODGAN_ACT_LRELU, "modconv_f16: act must be none or lrelu");
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
    static const int force_bpc = getenv("OODGAN_F16_BLOCKS_PER_CU") ? atoi(getenv("OODGAN_F16_BLOCKS_PER_CU")) : 0;
    const int KC = (K + 15) / 16;
#define OODGAN_LAUNCH(KC_)                                                                                              \
    {                                                                                                                   \
        constexpr int sm = 2 * Geo<KC_>::buf_bytes;                                                                     \
        static int occ = 0;                                                                                             \
        if (!occ) {                                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_f16_kernel<KC_>),                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, sm);                                  \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, modconv_f16_kernel<KC_>, 256, sm) != hipSuccess || occ < 1) \
                occ = 1;                                                                                                \
        }                                                                                                               \
        long g = (long)num_cu * (force_bpc ? force_bpc : occ);                                                          \
        g = (g + 7) / 8 * 8;                                                                                            \
        if (g > T) g = T;                                                                                               \
        hipLaunchKernelGGL((modconv_f16_kernel<KC_>), dim3((unsigned)g), dim3(256), sm, as_stream(stream), p);          \
    }
    if (KC == 2) OODGAN_LAUNCH(2) else OODGAN_LAUNCH(1)
#undef OODGAN_LAUNCH
    return check_launch("modconv_f16");
}
