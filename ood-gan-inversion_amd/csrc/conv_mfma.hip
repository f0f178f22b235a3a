// Implicit-GEMM 3x3 convolutions on the CDNA4 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// GEMM view:  Y[m, n] = sum_k  Wpk[k, tap, m] * X~[k, n + tap]      m = output channel
//                                                                    n = pixel of the N-space
//                                                                    k = input channel
// Three geometries share one kernel body:
//   S1  3x3 stride 1 pad 1           — plain ModulatedConv2d, its input gradient, AlignNet convs
//   T2  3x3 transposed stride 2      — the up-sampling ModulatedConv2d before its blur
//                                      (reference src/ops/StyleGAN/model.py:247-258)
//   S2  3x3 stride 2, no pad         — input gradient of T2
// Modulation is applied on the INPUT side while staging (x*style) and demodulation on the OUTPUT
// side in the epilogue, so the weights are shared by the whole batch and stay L2-resident
// (algebraically identical to the reference's B materialised weights: SURVEY.md Appendix A/E).
//
// Block = 4 waves (256 threads).  Per 32-lane half of a wave one MFMA k-slot: lanes 0-31 feed input
// channel c, lanes 32-63 channel c+1, so a single v_mfma_f32_32x32x2_f32 contracts two channels of one
// filter tap for 32 output channels x 32 pixels.  Tiles are staged through LDS:
//   x tile  [CK][IN_R][IN_C]   (halo included, zero filled, modulated)     — conflict-free ds_read_b32:
//                               the 32 lanes of a half read 32 consecutive pixels of one row
//   w tile  [CK][9][MB]        (MB = 32*MT output channels, channel fastest) — same property
// Global->register loads of chunk t+1 are issued before the MFMA loop of chunk t (register
// double-buffering, one LDS buffer, two barriers per chunk); 2+ blocks per CU hide the rest.
#include "conv_common.hpp"

using namespace oodgan;

namespace {

template <int MODE> struct Geo;
template <> struct Geo<OODGAN_CONV_S1> { static constexpr int TR = 8, NT = 2, IN_R = 10, IN_C = 34, CK = 8, NPH = 1; };
template <> struct Geo<OODGAN_CONV_T2> { static constexpr int TR = 4, NT = 1, IN_R = 5, IN_C = 33, CK = 8, NPH = 4; };
template <> struct Geo<OODGAN_CONV_S2> { static constexpr int TR = 8, NT = 2, IN_R = 17, IN_C = 66, CK = 4, NPH = 1; };

template <int MODE, int MT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const KArgs p) {
    using G = Geo<MODE>;
    constexpr int CK = G::CK, NT = G::NT, TR = G::TR, IN_R = G::IN_R, IN_C = G::IN_C, NPH = G::NPH;
    constexpr int XT = IN_R * IN_C;
    constexpr int MB = 32 * MT;
    constexpr int XE = CK * XT;
    constexpr int XPT = (XE + 255) / 256;
    constexpr int WROW = MB / 4;             // float4 per (c,tap) row
    constexpr int WE = CK * 9 * WROW;
    constexpr int WPT = (WE + 255) / 256;
    constexpr int NACC = (MODE == OODGAN_CONV_T2) ? 4 : NT;

    __shared__ __attribute__((aligned(16))) float lds[XE + CK * 9 * MB];
    float* lx = lds;
    float* lw = lds + XE;

    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    const BlockCtx ctx = decode_block<TR, MB>(p);
    const int b = ctx.b, r0 = ctx.r0, c0 = ctx.c0, m0 = ctx.m0;

    const float* xb = a.x + (long)b * a.K * p.in_plane;
    const float* isc = a.in_scale ? a.in_scale + (long)b * a.in_scale_stride : nullptr;
    const float* ish = a.in_shift ? a.in_shift + (long)b * a.in_scale_stride : nullptr;

    // ---- per-thread staging descriptors (chunk independent) ----
    int xoff[XPT];   // global offset inside one channel plane, or -1 if out of bounds
    int xch[XPT];    // channel within chunk
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int e = tid + i * 256;
        int c = e / XT, rem = e % XT;
        int r = rem / IN_C, col = rem % IN_C;
        int gy, gx;
        if (MODE == OODGAN_CONV_S2) {
            const int par = col / 33, idx = col % 33;
            gy = 2 * r0 + r;
            gx = 2 * (c0 + idx) + par;
        } else {
            gy = r0 - 1 + r;
            gx = c0 - 1 + col;
        }
        const bool ok = (e < XE) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        xoff[i] = ok ? gy * a.in_pitch + gx : -1;
        xch[i] = c;
    }

    float xr[XPT];
    float4 wr[WPT];

    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int k = k0 + xch[i];
            float v = 0.f;
            if (xoff[i] >= 0 && k < a.K) {
                v = xb[(long)k * p.in_plane + xoff[i]];
                if (isc) v *= isc[k];
                if (ish) v += ish[k];
            }
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;
            const int row = e / WROW, q = e % WROW;   // row = c*9 + tap
            const int k = k0 + row / 9;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < WE && k < a.K)
                v = *reinterpret_cast<const float4*>(a.wpk + ((long)k * 9 + (row % 9)) * p.Mp + m0 + q * 4);
            wr[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int e = tid + i * 256;
            if (e < XE) lx[e] = xr[i];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * 256;
            if (e < WE) reinterpret_cast<float4*>(lw)[e] = wr[i];
        }
    };

    f32x16 acc[MT][NACC];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][j][r] = 0.f;

    // lane bases (all further offsets are compile-time immediates)
    const float* lwh = lw + half * 9 * MB + l31;
    const float* lxh;
    if (MODE == OODGAN_CONV_S1) lxh = lx + half * XT + (wave * NT) * IN_C + l31;
    else if (MODE == OODGAN_CONV_T2) lxh = lx + half * XT + wave * IN_C + l31;
    else lxh = lx + half * XT + (wave * NT) * 2 * IN_C + l31;

    const int nchunk = (a.K + CK - 1) / CK;
    load_chunk(0);
    for (int t = 0; t < nchunk; ++t) {
        __syncthreads();          // previous chunk fully consumed
        store_chunk();
        __syncthreads();
        if (t + 1 < nchunk) load_chunk((t + 1) * CK);   // in flight during the MFMA loop

#pragma unroll
        for (int cp = 0; cp < CK / 2; ++cp) {
            const float* xw = lxh + cp * 2 * XT;
            const float* ww = lwh + cp * 2 * 9 * MB;
            if constexpr (MODE == OODGAN_CONV_T2) {
                float bs[2][2];
#pragma unroll
                for (int da = 0; da < 2; ++da)
#pragma unroll
                    for (int db = 0; db < 2; ++db) bs[da][db] = xw[(1 - da) * IN_C + (1 - db)];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float wv[9];
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) wv[tp] = ww[tp * MB + mt * 32];
                    // phase (py,px) = parity of (ky,kx); z[2i'+ky', 2j'+kx'] with x[i'-ky/2, j'-kx/2]
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[0], bs[0][0], acc[mt][0], 0, 0, 0);
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[2], bs[0][1], acc[mt][0], 0, 0, 0);
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[6], bs[1][0], acc[mt][0], 0, 0, 0);
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[8], bs[1][1], acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[1], bs[0][0], acc[mt][1], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[7], bs[1][0], acc[mt][1], 0, 0, 0);
                    acc[mt][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[3], bs[0][0], acc[mt][2], 0, 0, 0);
                    acc[mt][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[5], bs[0][1], acc[mt][2], 0, 0, 0);
                    acc[mt][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[4], bs[0][0], acc[mt][3], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float av[MT], bv[NT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) av[mt] = ww[(ky * 3 + kx) * MB + mt * 32];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if constexpr (MODE == OODGAN_CONV_S1) bv[nt] = xw[(nt + ky) * IN_C + kx];
                            else bv[nt] = xw[(2 * nt + ky) * IN_C + (kx == 1 ? 33 : (kx == 2 ? 1 : 0))];
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
                    }
            }
        }
    }

    conv_epilogue<MODE, MT, NT, NACC>(p, acc, ctx, wave, l31, half);
}

template <int MODE>
int launch_mode(const oodgan_conv_args& a, hipStream_t st) {
    using G = Geo<MODE>;
    KArgs p;
    p.a = a;
    p.w_unscale = nullptr;
    p.ablate = 0;
    if (MODE == OODGAN_CONV_S1) { p.Hn = a.Hin; p.Wn = a.Win; p.Hout = a.Hin; p.Wout = a.Win; }
    else if (MODE == OODGAN_CONV_T2) { p.Hn = a.Hin + 1; p.Wn = a.Win + 1; p.Hout = 2 * a.Hin + 1; p.Wout = 2 * a.Win + 1; }
    else { p.Hn = (a.Hin - 1) / 2; p.Wn = (a.Win - 1) / 2; p.Hout = p.Hn; p.Wout = p.Wn; }
    if (p.a.in_pitch == 0) p.a.in_pitch = a.Win;
    if (p.a.out_pitch == 0) p.a.out_pitch = p.Wout;
    p.in_plane = (long)a.Hin * p.a.in_pitch;
    p.out_plane = (long)p.Hout * p.a.out_pitch;
    if (MODE == OODGAN_CONV_T2) {
        OODGAN_REQUIRE((p.a.out_pitch & 1) == 0, "conv3x3 T2: out_pitch must be even (got %d)", p.a.out_pitch);
        if (p.out_plane & 1) p.out_plane += p.a.out_pitch;  // never hit: pitch even => plane even
        OODGAN_REQUIRE(a.dotx == nullptr && a.noise == nullptr && a.bias == nullptr && a.act == OODGAN_ACT_NONE,
                       "conv3x3 T2: only out_scale is supported in the epilogue");
    }
    p.tiles_y = (p.Hn + G::TR - 1) / G::TR;
    p.tiles_x = (p.Wn + 31) / 32;
    p.Mp = (a.M + 63) / 64 * 64;
    const bool mt2 = a.M > 32;
    const int MB = mt2 ? 64 : 32;
    p.mblocks = (a.M + MB - 1) / MB;
    if (a.dotx) {
        OODGAN_REQUIRE(a.dot_part != nullptr, "conv3x3: dotx without dot_part");
        OODGAN_REQUIRE(a.dot_nparts == p.tiles_x * p.tiles_y * 4, "conv3x3: dot_nparts %d != %d", a.dot_nparts,
                       p.tiles_x * p.tiles_y * 4);
    }
    const long total = (long)p.tiles_x * p.tiles_y * a.B * p.mblocks;
    OODGAN_REQUIRE(total > 0 && total < (1L << 31), "conv3x3: grid too large");
    dim3 grid((unsigned)total), block(256);
    if (mt2) hipLaunchKernelGGL((conv_mfma_kernel<MODE, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_mfma_kernel<MODE, 1>), grid, block, 0, st, p);
    return check_launch("conv3x3");
}

// wpk[k][tap][Mp]
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Co, int Ci, int Mp,
                                    float scale, int transpose, int flip) {
    const int K = transpose ? Co : Ci;
    const long total = (long)K * 9 * Mp;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int m = (int)(e % Mp);
        const int tap = (int)((e / Mp) % 9);
        const int k = (int)(e / ((long)Mp * 9));
        const int M = transpose ? Ci : Co;
        float v = 0.f;
        if (m < M) {
            const int co = transpose ? k : m, ci = transpose ? m : k;
            const int t = flip ? 8 - tap : tap;
            v = w[((long)co * Ci + ci) * 9 + t] * scale;
        }
        wpk[e] = v;
    }
}

}  // namespace

extern "C" int oodgan_conv3x3_nparts(int mode, int Hin, int Win) {
    if (mode == OODGAN_CONV_S1) return ((Hin + 7) / 8) * ((Win + 31) / 32) * 4;
    if (mode == OODGAN_CONV_S2) return (((Hin - 1) / 2 + 7) / 8) * (((Win - 1) / 2 + 31) / 32) * 4;
    return 0;
}

extern "C" int oodgan_conv3x3(const oodgan_conv_args* args, void* stream) {
    OODGAN_REQUIRE(args != nullptr, "conv3x3: null args");
    const oodgan_conv_args& a = *args;
    OODGAN_REQUIRE(a.x && a.wpk && a.y, "conv3x3: null tensor");
    if (!oodgan::bound_device_ok("conv3x3")) return OODGAN_E_ARG;
    OODGAN_REQUIRE(a.rgb_y == nullptr, "conv3x3: the fused ToRGB output exists only in the split-f16 strip kernel");
    OODGAN_REQUIRE(a.fuse == nullptr, "conv3x3: the fused activation backward exists only in the split-f16 stride-2 kernel");
    OODGAN_REQUIRE(!a.dot_actgrad, "conv3x3: dot_actgrad exists only in the split-f16 stride-1 kernels");
    OODGAN_REQUIRE(a.groups <= 1, "conv3x3: grouped convolution exists only in the split-f16 stride-2 kernel");
    OODGAN_REQUIRE(!a.y_fform, "conv3x3: the F-form output exists only in the split-f16 strip kernel");
    OODGAN_REQUIRE(a.B > 0 && a.K > 0 && a.M > 0 && a.Hin > 0 && a.Win > 0, "conv3x3: bad shape B=%d K=%d M=%d H=%d W=%d",
                   a.B, a.K, a.M, a.Hin, a.Win);
    OODGAN_REQUIRE(a.act != OODGAN_ACT_PRELU || a.slope, "conv3x3: PReLU without slopes");
    OODGAN_REQUIRE(a.noise == nullptr || a.noise_batch == 1 || a.noise_batch == a.B, "conv3x3: noise_batch");
    hipStream_t st = as_stream(stream);
    switch (a.mode) {
        case OODGAN_CONV_S1: return launch_mode<OODGAN_CONV_S1>(a, st);
        case OODGAN_CONV_T2: return launch_mode<OODGAN_CONV_T2>(a, st);
        case OODGAN_CONV_S2:
            OODGAN_REQUIRE((a.Hin & 1) && (a.Win & 1) && a.Hin >= 3 && a.Win >= 3, "conv3x3 S2: input must be odd-sized");
            return launch_mode<OODGAN_CONV_S2>(a, st);
        default: break;
    }
    set_error("conv3x3: unknown mode %d", a.mode);
    return OODGAN_E_ARG;
}

extern "C" int oodgan_pack_conv3x3(const float* w, float* wpk, int Co, int Ci, float scale, int transpose, int flip,
                                   void* stream) {
    OODGAN_REQUIRE(w && wpk && Co > 0 && Ci > 0, "pack_conv3x3: bad args");
    const int M = transpose ? Ci : Co, K = transpose ? Co : Ci;
    const int Mp = (M + 63) / 64 * 64;
    const long total = (long)K * 9 * Mp;
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), w, wpk, Co, Ci,
                       Mp, scale, transpose, flip);
    return check_launch("pack_conv3x3");
}
