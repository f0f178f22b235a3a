// upfirdn2d: zero-stuff (up) -> pad/crop -> FIR with the flipped kernel -> decimate (down).
// Semantics = reference upfirdn2d_native (src/ops/op/upfirdn2d.py:160-193); the CUDA kernel the
// reference ships (src/ops/op/upfirdn2d_kernel.cu:52-137) is unreachable on its default path and
// was used only as a statement of intent (tile + small FIR).
//
// Two kernels:
//   * fir_tile_kernel  — up=1, down=1 (the blur after the transposed conv and the SAMM flow blur): a
//     32x64 output tile + halo is staged in LDS once (each input element is read from HBM once) and
//     every lane produces 4 horizontally adjacent outputs from registers; optional fused
//     noise + bias + LeakyReLU*sqrt2 epilogue so the StyledConv tail costs no extra pass.
//   * generic_kernel   — any up/down/pad (3-channel skip up-sampling and its adjoint, tests).
#include "common.hpp"

using namespace oodgan;

namespace {

struct UfdArgs {
    const float* x;
    const float* k;
    float* y;
    const float* bias;      // (C) or null
    const float* noise;     // (noise_batch,1,out_h,out_w) or null
    const float* noise_w;
    int planes, C;          // C = channels per batch item (for bias / noise indexing)
    int in_h, in_w, in_pitch;
    int out_h, out_w, out_pitch;
    int kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0;
    int noise_batch, act;
};

__global__ __launch_bounds__(256) void generic_kernel(const UfdArgs a) {
    const long total = (long)a.planes * a.out_h * a.out_w;
    const long in_plane = (long)a.in_h * a.in_pitch, out_plane = (long)a.out_h * a.out_pitch;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(e % a.out_w);
        const int oy = (int)((e / a.out_w) % a.out_h);
        const long pl = e / ((long)a.out_w * a.out_h);
        const float* xp = a.x + pl * in_plane;
        float acc = 0.f;
        for (int ky = 0; ky < a.kh; ++ky) {
            const int u = oy * a.down_y + ky - a.pad_y0;     // position on the zero-stuffed grid
            if (u < 0 || u % a.up_y) continue;
            const int iy = u / a.up_y;
            if (iy >= a.in_h) continue;
            for (int kx = 0; kx < a.kw; ++kx) {
                const int v = ox * a.down_x + kx - a.pad_x0;
                if (v < 0 || v % a.up_x) continue;
                const int ix = v / a.up_x;
                if (ix >= a.in_w) continue;
                acc += a.k[(a.kh - 1 - ky) * a.kw + (a.kw - 1 - kx)] * xp[(long)iy * a.in_pitch + ix];
            }
        }
        if (a.noise) {
            const long b = pl / a.C;
            acc += (a.noise_w ? a.noise_w[0] : 1.f) * a.noise[(a.noise_batch > 1 ? b : 0) * (long)a.out_h * a.out_w + (long)oy * a.out_w + ox];
        }
        if (a.bias) acc += a.bias[pl % a.C];
        if (a.act == OODGAN_ACT_LRELU) acc = (acc > 0.f ? acc : 0.2f * acc) * kSqrt2;
        a.y[pl * out_plane + (long)oy * a.out_pitch + ox] = acc;
    }
}

// up=1, down=2, 4x4 kernel (adjoint of the ToRGB skip up-sampling: the gradient chain gskip[r] -> gskip[r/2] of the W+
// loop): the generic kernel's per-tap division / modulo / bounds branches made the 1024² -> 512² instance run at
// 0.66 TB/s; here the 16 taps are unrolled, interior outputs take a branch-free path.
__global__ __launch_bounds__(256) void down2_k4_kernel(const UfdArgs a) {
    __shared__ float kf[16];
    if (threadIdx.x < 16) kf[threadIdx.x] = a.k[15 - threadIdx.x];       // flipped: kf[ky][kx] = k[3-ky][3-kx]
    __syncthreads();
    const long total = (long)a.planes * a.out_h * a.out_w;
    const long in_plane = (long)a.in_h * a.in_pitch, out_plane = (long)a.out_h * a.out_pitch;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % a.out_w);
        const int oy = (int)((e / a.out_w) % a.out_h);
        const long pl = e / ((long)a.out_w * a.out_h);
        const float* xp = a.x + pl * in_plane;
        const int y0 = 2 * oy - a.pad_y0, x0 = 2 * ox - a.pad_x0;
        // the 16 taps from clamped positions, all in flight together, a tap outside the image masked by a select (a conditional load
        // per tap is a round trip per tap; the launches of this kernel are latency-bound: 3 x B planes)
        float v[16];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
                v[ky * 4 + kx] = xp[(long)min(max(y0 + ky, 0), a.in_h - 1) * a.in_pitch + min(max(x0 + kx, 0), a.in_w - 1)];
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int iy = y0 + ky, ix = x0 + kx;
                if (iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) acc += kf[ky * 4 + kx] * v[ky * 4 + kx];
            }
        a.y[pl * out_plane + (long)oy * a.out_pitch + ox] = acc;
    }
}

// up=1, down=1, kh,kw <= 4.  Output tile 32 rows x 64 cols per block; thread (ty 0..15, tx 0..15)
// produces rows {ty, ty+16} x cols 4*tx..4*tx+3.
constexpr int FT_H = 32, FT_W = 64, FK = 4;
constexpr int FL_H = FT_H + FK - 1, FL_W = FT_W + FK - 1;   // 35 x 67
constexpr int FL_P = FL_W + 1;                               // LDS pitch 68

__global__ __launch_bounds__(256) void fir_tile_kernel(const UfdArgs a, int tiles_x, int tiles_y) {
    __shared__ float t[FL_H * FL_P];
    __shared__ float kf[FK * FK];
    int w = blockIdx.x;
    const int tx_ = w % tiles_x; w /= tiles_x;
    const int ty_ = w % tiles_y; w /= tiles_y;
    const long pl = w;
    const int oy0 = ty_ * FT_H, ox0 = tx_ * FT_W;
    const long in_plane = (long)a.in_h * a.in_pitch, out_plane = (long)a.out_h * a.out_pitch;
    const float* xp = a.x + pl * in_plane;
    if (threadIdx.x < FK * FK) {
        const int ky = threadIdx.x / FK, kx = threadIdx.x % FK;
        kf[threadIdx.x] = (ky < a.kh && kx < a.kw) ? a.k[(a.kh - 1 - ky) * a.kw + (a.kw - 1 - kx)] : 0.f;
    }
    const int iy0 = oy0 - a.pad_y0, ix0 = ox0 - a.pad_x0;
    {
        // the whole fill of the tile in one request: every element from a clamped position, masked when it goes to LDS (a conditional
        // load per trip, stored right away, is a round trip per trip: ten in a row)
        constexpr int NF = (FL_H * FL_W + 255) / 256;
        float fv[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int e = min((int)threadIdx.x + 256 * i, FL_H * FL_W - 1);
            const int iy = iy0 + e / FL_W, ix = ix0 + e % FL_W;
            fv[i] = xp[(long)min(max(iy, 0), a.in_h - 1) * a.in_pitch + min(max(ix, 0), a.in_w - 1)];
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int e = (int)threadIdx.x + 256 * i;
            if (e >= FL_H * FL_W) continue;
            const int r = e / FL_W, c = e % FL_W;
            const int iy = iy0 + r, ix = ix0 + c;
            t[r * FL_P + c] = (iy >= 0 && iy < a.in_h && ix >= 0 && ix < a.in_w) ? fv[i] : 0.f;
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float kr[FK * FK];
#pragma unroll
    for (int i = 0; i < FK * FK; ++i) kr[i] = kf[i];
    const long b = pl / a.C;
    const float bv = a.bias ? a.bias[pl % a.C] : 0.f;
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const float* np = a.noise ? a.noise + (a.noise_batch > 1 ? b : 0) * (long)a.out_h * a.out_w : nullptr;
    // the noise of this thread's eight outputs, requested before the filter runs (clamped pixel; the input plane as a valid address
    // when there is no noise)
    float nzv[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = min(oy0 + ty + rr * 16, a.out_h - 1), ox = min(ox0 + 4 * tx + j, a.out_w - 1);
            nzv[rr][j] = *(np ? np + (long)oy * a.out_w + ox : xp);
        }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = ty + rr * 16;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < FK; ++ky) {
            float row[4 + FK - 1];
#pragma unroll
            for (int j = 0; j < 4 + FK - 1; ++j) row[j] = t[(r + ky) * FL_P + 4 * tx + j];
#pragma unroll
            for (int kx = 0; kx < FK; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += kr[ky * FK + kx] * row[j + kx];
        }
        const int oy = oy0 + r;
        if (oy >= a.out_h) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = ox0 + 4 * tx + j;
            if (ox >= a.out_w) continue;
            float v = o[j];
            if (np) v += nw * nzv[rr][j];
            v += bv;
            if (a.act == OODGAN_ACT_LRELU) v = (v > 0.f ? v : 0.2f * v) * kSqrt2;
            a.y[pl * out_plane + (long)oy * a.out_pitch + ox] = v;
        }
    }
}

int run(UfdArgs& a, int pad_x1, int pad_y1, hipStream_t st) {
    OODGAN_REQUIRE(a.x && a.k && a.y, "upfirdn2d: null tensor");
    OODGAN_REQUIRE(a.planes > 0 && a.in_h > 0 && a.in_w > 0 && a.kh > 0 && a.kw > 0, "upfirdn2d: bad shape");
    OODGAN_REQUIRE(a.up_x >= 1 && a.up_y >= 1 && a.down_x >= 1 && a.down_y >= 1, "upfirdn2d: up/down must be >= 1");
    a.out_h = (a.in_h * a.up_y + a.pad_y0 + pad_y1 - a.kh) / a.down_y + 1;
    a.out_w = (a.in_w * a.up_x + a.pad_x0 + pad_x1 - a.kw) / a.down_x + 1;
    OODGAN_REQUIRE(a.out_h > 0 && a.out_w > 0, "upfirdn2d: empty output (%d x %d)", a.out_h, a.out_w);
    if (a.in_pitch == 0) a.in_pitch = a.in_w;
    if (a.out_pitch == 0) a.out_pitch = a.out_w;
    OODGAN_REQUIRE(a.in_pitch >= a.in_w && a.out_pitch >= a.out_w, "upfirdn2d: pitch smaller than width");
    const bool tiled = a.up_x == 1 && a.up_y == 1 && a.down_x == 1 && a.down_y == 1 && a.kh <= FK && a.kw <= FK &&
                       (long)a.out_h * a.out_w >= 64 * 64;
    if (tiled) {
        const int tiles_x = (a.out_w + FT_W - 1) / FT_W, tiles_y = (a.out_h + FT_H - 1) / FT_H;
        const long nb = (long)tiles_x * tiles_y * a.planes;
        OODGAN_REQUIRE(nb < (1L << 31), "upfirdn2d: grid too large");
        hipLaunchKernelGGL(fir_tile_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, tiles_x, tiles_y);
    } else if (a.up_x == 1 && a.up_y == 1 && a.down_x == 2 && a.down_y == 2 && a.kh == 4 && a.kw == 4 && !a.noise && !a.bias &&
               a.act == OODGAN_ACT_NONE) {
        const long total = (long)a.planes * a.out_h * a.out_w;
        hipLaunchKernelGGL(down2_k4_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, a);
    } else {
        const long total = (long)a.planes * a.out_h * a.out_w;
        hipLaunchKernelGGL(generic_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, a);
    }
    return check_launch("upfirdn2d");
}

}  // namespace

extern "C" int oodgan_upfirdn2d(const float* x, const float* kernel, float* y, int planes, int in_h, int in_w, int in_pitch,
                                int out_pitch, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                                int pad_x1, int pad_y0, int pad_y1, void* stream) {
    UfdArgs a{};
    a.x = x; a.k = kernel; a.y = y;
    a.planes = planes; a.C = 1;
    a.in_h = in_h; a.in_w = in_w; a.in_pitch = in_pitch; a.out_pitch = out_pitch;
    a.kh = kh; a.kw = kw; a.up_x = up_x; a.up_y = up_y; a.down_x = down_x; a.down_y = down_y;
    a.pad_x0 = pad_x0; a.pad_y0 = pad_y0;
    a.act = OODGAN_ACT_NONE; a.noise_batch = 1;
    return run(a, pad_x1, pad_y1, as_stream(stream));
}

extern "C" int oodgan_blur_bias_act(const float* x, const float* kernel, float* y, int B, int C, int in_h, int in_w,
                                    int in_pitch, int kh, int kw, int pad0, int pad1, const float* bias, const float* noise,
                                    int noise_batch, const float* noise_w, int act, void* stream) {
    OODGAN_REQUIRE(B > 0 && C > 0, "blur_bias_act: bad batch/channels");
    OODGAN_REQUIRE(noise == nullptr || noise_batch == 1 || noise_batch == B, "blur_bias_act: noise_batch");
    UfdArgs a{};
    a.x = x; a.k = kernel; a.y = y;
    a.planes = B * C; a.C = C;
    a.in_h = in_h; a.in_w = in_w; a.in_pitch = in_pitch; a.out_pitch = 0;
    a.kh = kh; a.kw = kw; a.up_x = a.up_y = a.down_x = a.down_y = 1;
    a.pad_x0 = pad0; a.pad_y0 = pad0;
    a.bias = bias; a.noise = noise; a.noise_w = noise_w; a.noise_batch = noise_batch; a.act = act;
    return run(a, pad1, pad1, as_stream(stream));
}
