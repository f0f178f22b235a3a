// Style path: the 26 style affines of a generator forward as ONE dense contraction on the fp32 matrix
// cores, demodulation coefficients by wavefront reduction, the mapping network layers.
//   reference: EqualLinear (src/ops/StyleGAN/model.py:129-158), ModulatedConv2d.forward :236-241.
#include "common.hpp"

using namespace oodgan;

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// s[b, r] = scale * sum_k W[r,k] * lat[b, row_lat[r], k] + bias[r]*lr_mul
// MFMA path: one wave per 16-row tile x 16 batch columns; v_mfma_f32_16x16x4_f32:
//   A[i=l&15][k=l>>4] = W[r0+i][k], B[k=l>>4][j=l&15] = lat[b0+j][.][k]; each lane loads 16 B of both
//   operands per 16-deep K step and issues 4 MFMAs (k order inside a step is irrelevant for a sum as long
//   as A and B agree).  Requires all 16 rows of a tile to read the same latent row.
__global__ __launch_bounds__(256) void style_affine_mfma_kernel(const float* __restrict__ lat, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const int* __restrict__ row_lat,
                                                                float* __restrict__ s, int B, int L, int S, int R, float scale,
                                                                float lr_mul) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = tile * 16;
    if (r0 >= R) return;
    const int b0 = blockIdx.y * 16;
    const int i = lane & 15, kq = lane >> 4;
    const int li = row_lat ? row_lat[r0] : 0;
    const int bj = b0 + i;
    const float* wp = w + (long)(r0 + i) * S + 4 * kq;
    const float* lp = lat + ((long)(bj < B ? bj : 0) * L + li) * S + 4 * kq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < S; k0 += 16) {
        const float4 av = *reinterpret_cast<const float4*>(wp + k0);
        float4 bv = *reinterpret_cast<const float4*>(lp + k0);
        if (bj >= B) bv = make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
    }
    // C/D: col = lane&15 (batch), row = (lane>>4)*4 + reg
    if (bj < B) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + kq * 4 + q;
            s[(long)bj * R + r] = acc[q] * scale + (bias ? bias[r] * lr_mul : 0.f);
        }
    }
}

// generic path (any R, per-row latent index): one wave per row
__global__ __launch_bounds__(256) void style_affine_wave_kernel(const float* __restrict__ lat, const float* __restrict__ w,
                                                                const float* __restrict__ bias, const int* __restrict__ row_lat,
                                                                float* __restrict__ s, int B, int L, int S, int R, float scale,
                                                                float lr_mul) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int li = row_lat ? row_lat[r] : 0;
    const float bv = bias ? bias[r] * lr_mul : 0.f;
    for (int b = 0; b < B; ++b) {
        const float* lp = lat + ((long)b * L + li) * S;
        float acc = 0.f;
        for (int k = lane; k < S; k += 64) acc += w[r * S + k] * lp[k];
        acc = wave_sum(acc);
        if (lane == 0) s[(long)b * R + r] = acc * scale + bv;
    }
}

// glat[b,l,k] = scale * sum_{r in [lat_start[l], lat_start[l+1])} gs[b,r] * W[r,k]
// grid (ceil(S/64), L, ceil(B/8)); block = 64 k-columns x 16 row groups (rows of the latent split 16 ways, reduced
// through LDS in a fixed order); W row reads are 256-B coalesced, the 8 batch accumulators share every W element.
// (With 4 row groups the ~500 rows of a latent were a chain of 125 dependent-latency iterations per thread: 325 us.)
constexpr int SAB_RG = 16, SAB_CHUNK = 1024;
__global__ __launch_bounds__(64 * SAB_RG) void style_affine_bwd_kernel(const float* __restrict__ gs, const float* __restrict__ w,
                                                                      const int* __restrict__ lat_start, float* __restrict__ glat,
                                                                      int B, int L, int S, int R, float scale) {
    __shared__ float red[SAB_RG][8][64];
    __shared__ float gsl[8][SAB_CHUNK];
    const int kl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    const int l = blockIdx.y;
    const int b0 = blockIdx.z * 8;
    const int ra = lat_start[l], rb = lat_start[l + 1];
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // the latent's slice of gs (8 batch rows) is staged in LDS: read from global it is 8 broadcast loads per W element
    for (int c0 = ra; c0 < rb; c0 += SAB_CHUNK) {
        const int cn = rb - c0 < SAB_CHUNK ? rb - c0 : SAB_CHUNK;
        __syncthreads();
        for (int e = threadIdx.x; e < 8 * cn; e += 64 * SAB_RG) {
            const int j = e / cn, r = e - j * cn;
            gsl[j][r] = b0 + j < B ? gs[(long)(b0 + j) * R + c0 + r] : 0.f;
        }
        __syncthreads();
        if (k < S) {
#pragma unroll 4
            for (int r = rg; r < cn; r += SAB_RG) {
                const float wv = w[(long)(c0 + r) * S + k];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += gsl[j][r] * wv;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rg][j][kl] = acc[j];
    __syncthreads();
    if (rg < 8 && k < S) {          // wave j sums batch row j over the row groups, in a fixed order
        const int b = b0 + rg;
        if (b < B) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < SAB_RG; ++g) v += red[g][rg][kl];
            glat[((long)b * L + l) * S + k] = v * scale;
        }
    }
}

__global__ __launch_bounds__(256) void equal_linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B,
                                                           int in_dim, int out_dim, float scale, float lr_mul, int activate) {
    const int lane = threadIdx.x & 63;
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (long)B * out_dim) return;
    const int b = (int)(idx / out_dim), o = (int)(idx % out_dim);
    float acc = 0.f;
    for (int k = lane; k < in_dim; k += 64) acc += x[(long)b * in_dim + k] * (w[(long)o * in_dim + k] * scale);
    acc = wave_sum(acc);
    if (lane == 0) {
        float v = acc + (bias ? bias[o] * lr_mul : 0.f);
        if (activate) v = (v > 0.f ? v : 0.2f * v) * kSqrt2;
        y[idx] = v;
    }
}

// G independent EqualLinear layers side by side (the final linears of the e4e style heads): x (B,G,I), w (G,O,I), bias (G,O) -> y (B,G,O);
// one wave per output, the arithmetic and summation order of equal_linear_kernel (bit-identical to G separate calls)
__global__ __launch_bounds__(256) void equal_linear_grouped_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float* __restrict__ y, int B, int G,
                                                                   int in_dim, int out_dim, float scale, float lr_mul, int activate) {
    const int lane = threadIdx.x & 63;
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (long)B * G * out_dim) return;
    const int o = (int)(idx % out_dim), g = (int)((idx / out_dim) % G), b = (int)(idx / ((long)out_dim * G));
    const float* xp = x + ((long)b * G + g) * in_dim;
    const float* wp = w + ((long)g * out_dim + o) * in_dim;
    float acc = 0.f;
    for (int k = lane; k < in_dim; k += 64) acc += xp[k] * (wp[k] * scale);
    acc = wave_sum(acc);
    if (lane == 0) {
        float v = acc + (bias ? bias[(long)g * out_dim + o] * lr_mul : 0.f);
        if (activate) v = (v > 0.f ? v : 0.2f * v) * kSqrt2;
        y[idx] = v;
    }
}

__global__ __launch_bounds__(64) void pixel_norm_kernel(const float* __restrict__ x, float* __restrict__ y, int S) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int k = lane; k < S; k += 64) { const float v = x[(long)b * S + k]; acc += v * v; }
    acc = wave_sum(acc);
    const float r = rsqrtf(acc / (float)S + 1e-8f);
    for (int k = lane; k < S; k += 64) y[(long)b * S + k] = x[(long)b * S + k] * r;
}

__global__ __launch_bounds__(256) void weight_sqsum_kernel(const float* __restrict__ w, float* __restrict__ wsq, long n, int KK) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int k = 0; k < KK; ++k) { const float v = w[i * KK + k]; acc += v * v; }
    wsq[i] = acc;
}

// one wave per (b,co): d = rsqrt(scale^2 * sum_ci s^2 * wsq + 1e-8)   (literal 1e-8: model.py:240)
__global__ __launch_bounds__(256) void demod_fwd_kernel(const float* __restrict__ s, int s_stride, const float* __restrict__ wsq,
                                                        float* __restrict__ d, int d_stride, int B, int Ci, int Co, float scale2) {
    const int lane = threadIdx.x & 63;
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (long)B * Co) return;
    const int b = (int)(idx / Co), co = (int)(idx % Co);
    const float acc = demod_dot(s + (long)b * s_stride, wsq + (long)co * Ci, Ci, lane);       // common.hpp: shared with the batched kernel
    if (lane == 0) d[(long)b * d_stride + co] = rsqrtf(acc * scale2 + 1e-8f);
}

// gs[b,ci] += -scale^2 * s[b,ci] * sum_co r[b,co]*d[b,co]^2*wsq[co,ci]
// grid (ceil(Ci/64), B); block = 64 ci x 4 co-groups; r*d^2 staged in LDS, wsq rows read 256-B coalesced.
__global__ __launch_bounds__(256) void demod_bwd_kernel(const float* __restrict__ s, int s_stride, const float* __restrict__ wsq,
                                                        const float* __restrict__ d, int d_stride, const float* __restrict__ r,
                                                        float* __restrict__ gs, int gs_stride, int B, int Ci, int Co, float scale2) {
    __shared__ float rd2[1024];
    __shared__ float red[4][64];
    const int b = blockIdx.y;
    const int cil = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int ci = blockIdx.x * 64 + cil;
    float acc = 0.f;
    for (int c0 = 0; c0 < Co; c0 += 1024) {
        __syncthreads();
        for (int e = threadIdx.x; e < 1024 && c0 + e < Co; e += 256) {
            const float dv = d[(long)b * d_stride + c0 + e];
            rd2[e] = r[(long)b * Co + c0 + e] * dv * dv;
        }
        __syncthreads();
        const int cn = (Co - c0) < 1024 ? (Co - c0) : 1024;
        if (ci < Ci) {
#pragma unroll 8
            for (int co = cg; co < cn; co += 4) acc += rd2[co] * wsq[(long)(c0 + co) * Ci + ci];
        }
    }
    red[cg][cil] = acc;
    __syncthreads();
    if (cg == 0 && ci < Ci)
        gs[(long)b * gs_stride + ci] += -scale2 * s[(long)b * s_stride + ci] * (red[0][cil] + red[1][cil] + red[2][cil] + red[3][cil]);
}

}  // namespace

extern "C" int oodgan_style_affine_fwd(const float* latent, const float* wcat, const float* bcat, const int* row_lat, float* s,
                                       int B, int L, int S, int R, float scale, float lr_mul, void* stream) {
    OODGAN_REQUIRE(latent && wcat && s && B > 0 && L > 0 && S > 0 && R > 0, "style_affine_fwd: bad args");
    // negative S selects the generic path explicitly (per-row latent index not tile-uniform)
    hipStream_t st = as_stream(stream);
    const bool mfma = (R % 16 == 0) && (S % 16 == 0);
    if (mfma)
        hipLaunchKernelGGL(style_affine_mfma_kernel, dim3((R / 16 + 3) / 4, (B + 15) / 16), dim3(256), 0, st, latent, wcat, bcat,
                           row_lat, s, B, L, S, R, scale, lr_mul);
    else
        hipLaunchKernelGGL(style_affine_wave_kernel, dim3((R + 3) / 4), dim3(256), 0, st, latent, wcat, bcat, row_lat, s, B, L,
                           S, R, scale, lr_mul);
    return check_launch("style_affine_fwd");
}

extern "C" int oodgan_style_affine_bwd(const float* gs, const float* wcat, const int* lat_start, float* glat, int B, int L,
                                       int S, int R, float scale, void* stream) {
    OODGAN_REQUIRE(gs && wcat && lat_start && glat && B > 0 && L > 0 && S > 0 && R > 0, "style_affine_bwd: bad args");
    hipLaunchKernelGGL(style_affine_bwd_kernel, dim3((S + 63) / 64, L, (B + 7) / 8), dim3(64 * SAB_RG), 0, as_stream(stream), gs,
                       wcat, lat_start, glat, B, L, S, R, scale);
    return check_launch("style_affine_bwd");
}

extern "C" int oodgan_equal_linear(const float* x, const float* w, const float* b, float* y, int B, int in_dim, int out_dim,
                                   float scale, float lr_mul, int activate, void* stream) {
    OODGAN_REQUIRE(x && w && y && B > 0 && in_dim > 0 && out_dim > 0, "equal_linear: bad args");
    const long n = (long)B * out_dim;
    hipLaunchKernelGGL(equal_linear_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), x, w, b, y, B, in_dim,
                       out_dim, scale, lr_mul, activate);
    return check_launch("equal_linear");
}

extern "C" int oodgan_equal_linear_grouped(const float* x, const float* w, const float* b, float* y, int B, int G, int in_dim, int out_dim,
                                           float scale, float lr_mul, int activate, void* stream) {
    OODGAN_REQUIRE(x && w && y && B > 0 && G > 0 && in_dim > 0 && out_dim > 0, "equal_linear_grouped: bad args");
    const long n = (long)B * G * out_dim;
    hipLaunchKernelGGL(equal_linear_grouped_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), x, w, b, y, B, G, in_dim,
                       out_dim, scale, lr_mul, activate);
    return check_launch("equal_linear_grouped");
}

extern "C" int oodgan_pixel_norm(const float* x, float* y, int B, int S, void* stream) {
    OODGAN_REQUIRE(x && y && B > 0 && S > 0, "pixel_norm: bad args");
    hipLaunchKernelGGL(pixel_norm_kernel, dim3(B), dim3(64), 0, as_stream(stream), x, y, S);
    return check_launch("pixel_norm");
}

extern "C" int oodgan_weight_sqsum(const float* w, float* wsq, int Co, int Ci, int KK, void* stream) {
    OODGAN_REQUIRE(w && wsq && Co > 0 && Ci > 0 && KK > 0, "weight_sqsum: bad args");
    const long n = (long)Co * Ci;
    hipLaunchKernelGGL(weight_sqsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, wsq, n, KK);
    return check_launch("weight_sqsum");
}

extern "C" int oodgan_demod_fwd(const float* s, int s_stride, const float* wsq, float* d, int d_stride, int B, int Ci, int Co,
                                float scale, void* stream) {
    OODGAN_REQUIRE(s && wsq && d && B > 0 && Ci > 0 && Co > 0, "demod_fwd: bad args");
    const long n = (long)B * Co;
    hipLaunchKernelGGL(demod_fwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), s, s_stride, wsq, d,
                       d_stride, B, Ci, Co, scale * scale);
    return check_launch("demod_fwd");
}

extern "C" int oodgan_demod_bwd(const float* s, int s_stride, const float* wsq, const float* d, int d_stride, const float* r,
                                float* gs, int gs_stride, int B, int Ci, int Co, float scale, void* stream) {
    OODGAN_REQUIRE(s && wsq && d && r && gs && B > 0 && Ci > 0 && Co > 0, "demod_bwd: bad args");
    hipLaunchKernelGGL(demod_bwd_kernel, dim3((Ci + 63) / 64, B), dim3(256), 0, as_stream(stream), s, s_stride, wsq, d,
                       d_stride, r, gs, gs_stride, B, Ci, Co, scale * scale);
    return check_launch("demod_bwd");
}
