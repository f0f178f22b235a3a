// ToRGB: 1x1 modulated conv Ci->3 without demodulation + bias + 2x FIR-upsampled skip, one pass.
// reference: ToRGB.forward / Upsample (src/ops/StyleGAN/model.py:353-372, 30-48).
// HBM-bound (AI ~1.4 flop/B): every feature element is read exactly once with 16-B loads; the three
// modulated weight rows live in LDS; the skip (3 channels at half resolution) is gathered from L2.
#include "common.hpp"
#include "sform.hpp"

using namespace oodgan;

namespace {

// upfirdn2d(skip, k, up=2, pad=(2,1)) at pixel (Y, X): the 2 x 2 taps with (Y+ky-2), (X+kx-2) even.  fetch(): the 12 values from
// clamped positions, all in flight together; a tap outside the image gets a zero weight.  (A conditional load per tap is a round
// trip per tap: `if (inside) o += k * sp[q]` four times in a row.)  add(): same order of the sums as the conditional form.
struct SkipTaps {
    float v[4][3], w[4];
    template <class KF>
    __device__ __forceinline__ void fetch(const float* __restrict__ sp, int h2, int w2, int Y, int X, KF kf) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ky = (Y & 1) + 2 * t;
            const int iy = (Y + ky - 2) >> 1;
            const bool yok = Y + ky - 2 >= 0 && iy < h2;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kx = (X & 1) + 2 * u;
                const int ix = (X + kx - 2) >> 1;
                const bool ok = yok && X + kx - 2 >= 0 && ix < w2;
                const long q = (long)min(max(iy, 0), h2 - 1) * w2 + min(max(ix, 0), w2 - 1);
                w[2 * t + u] = ok ? kf(ky, kx) : 0.f;
                v[2 * t + u][0] = sp[q];
                v[2 * t + u][1] = sp[(long)h2 * w2 + q];
                v[2 * t + u][2] = sp[2L * h2 * w2 + q];
            }
        }
    }
    __device__ __forceinline__ void add(float& o0, float& o1, float& o2) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (w[k] != 0.f) { o0 += w[k] * v[k][0]; o1 += w[k] * v[k][1]; o2 += w[k] * v[k][2]; }
        }
    }
};


constexpr int kMaxCi = 1024;

// grid (ceil(HW/1024), B); thread = 4 consecutive pixels
__global__ __launch_bounds__(256) void torgb_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ s, int s_stride, const float* __restrict__ bias,
                                                        const float* __restrict__ skip, const float* __restrict__ kern,
                                                        float* __restrict__ y, int Ci, int H, int W, float scale) {
    __shared__ float ws[3 * kMaxCi];
    __shared__ float kf[16];
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    for (int e = threadIdx.x; e < 3 * Ci; e += 256) {
        const int c = e / Ci, ci = e % Ci;
        ws[e] = scale * w[c * Ci + ci] * s[(long)b * s_stride + ci];
    }
    if (threadIdx.x < 16 && skip) kf[threadIdx.x] = kern[(3 - threadIdx.x / 4) * 4 + (3 - threadIdx.x % 4)];  // flipped
    __syncthreads();
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    const float* xp = x + (long)b * Ci * HW + p;
    float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    const bool vec = (HW & 3) == 0;
    if (vec) {
#pragma unroll 8
        for (int ci = 0; ci < Ci; ++ci) {
            const float4 v = *reinterpret_cast<const float4*>(xp + (long)ci * HW);
            const float w0 = ws[ci], w1 = ws[Ci + ci], w2 = ws[2 * Ci + ci];
            a0[0] += w0 * v.x; a0[1] += w0 * v.y; a0[2] += w0 * v.z; a0[3] += w0 * v.w;
            a1[0] += w1 * v.x; a1[1] += w1 * v.y; a1[2] += w1 * v.z; a1[3] += w1 * v.w;
            a2[0] += w2 * v.x; a2[1] += w2 * v.y; a2[2] += w2 * v.z; a2[3] += w2 * v.w;
        }
    } else {
        for (int ci = 0; ci < Ci; ++ci) {
            const float w0 = ws[ci], w1 = ws[Ci + ci], w2 = ws[2 * Ci + ci];
            for (int j = 0; j < 4; ++j)
                if (p + j < HW) {
                    const float v = xp[(long)ci * HW + j];
                    a0[j] += w0 * v; a1[j] += w1 * v; a2[j] += w2 * v;
                }
        }
    }
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f, b2 = bias ? bias[2] : 0.f;
    const int h2 = H >> 1, w2_ = W >> 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (p + j >= HW) break;
        float o0 = a0[j] + b0, o1 = a1[j] + b1, o2 = a2[j] + b2;
        if (skip) {
            const int Y = (int)((p + j) / W), X = (int)((p + j) % W);
            SkipTaps st;
            st.fetch(skip + (long)b * 3 * h2 * w2_, h2, w2_, Y, X, [&](int ky, int kx) { return kf[ky * 4 + kx]; });
            st.add(o0, o1, o2);
        }
        float* yp = y + (long)b * 3 * HW + p + j;
        yp[0] = o0; yp[HW] = o1; yp[2 * HW] = o2;
    }
}

// Low resolutions (HW <= 4096): the per-thread loop over 512 input channels is a chain of dependent loads with only a
// handful of threads alive, so here ONE WAVE owns 4 consecutive pixels and its 64 lanes split the channels; the 12
// partial sums (3 colours x 4 pixels) are combined with a wave reduction.  grid = (HW/4 / 4 waves, B).
__global__ __launch_bounds__(256) void torgb_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ s, int s_stride,
                                                              const float* __restrict__ bias, const float* __restrict__ skip,
                                                              const float* __restrict__ kern, float* __restrict__ y, int Ci, int H,
                                                              int W, float scale) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long HW = (long)H * W;
    const long p = ((long)blockIdx.x * 4 + wave) * 4;
    if (p >= HW) return;
    const float* xp = x + (long)b * Ci * HW + p;
    float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int ci = lane; ci < Ci; ci += 64) {         // unrolled: the loads of four trips in flight (one trip was a bare round trip)
        const float4 v = *reinterpret_cast<const float4*>(xp + (long)ci * HW);
        const float sv = scale * s[(long)b * s_stride + ci];
        const float w0 = sv * w[ci], w1 = sv * w[Ci + ci], w2 = sv * w[2 * Ci + ci];
        a0[0] += w0 * v.x; a0[1] += w0 * v.y; a0[2] += w0 * v.z; a0[3] += w0 * v.w;
        a1[0] += w1 * v.x; a1[1] += w1 * v.y; a1[2] += w1 * v.z; a1[3] += w1 * v.w;
        a2[0] += w2 * v.x; a2[1] += w2 * v.y; a2[2] += w2 * v.z; a2[3] += w2 * v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a0[j] = wave_sum(a0[j]);
        a1[j] = wave_sum(a1[j]);
        a2[j] = wave_sum(a2[j]);
    }
    if (lane >= 4) return;
    const int j = lane;                       // lane j finishes pixel p + j
    float o0 = a0[0], o1 = a1[0], o2 = a2[0];
    if (j == 1) { o0 = a0[1]; o1 = a1[1]; o2 = a2[1]; }
    if (j == 2) { o0 = a0[2]; o1 = a1[2]; o2 = a2[2]; }
    if (j == 3) { o0 = a0[3]; o1 = a1[3]; o2 = a2[3]; }
    o0 += bias ? bias[0] : 0.f;
    o1 += bias ? bias[1] : 0.f;
    o2 += bias ? bias[2] : 0.f;
    if (skip) {
        const int h2 = H >> 1, w2_ = W >> 1;
        const int Y = (int)((p + j) / W), X = (int)((p + j) % W);
        SkipTaps st;
        st.fetch(skip + (long)b * 3 * h2 * w2_, h2, w2_, Y, X, [&](int ky, int kx) { return kern[(3 - ky) * 4 + (3 - kx)]; });      // flipped
        st.add(o0, o1, o2);
    }
    float* yp = y + (long)b * 3 * HW + p + j;
    yp[0] = o0; yp[HW] = o1; yp[2 * HW] = o2;
}

// ToRGB that ALSO emits the S-form input of the following up-sampling conv (x * its style, hi/lo split): the feature map
// is read once for both consumers instead of once by ToRGB and once by to_sform_kernel.  Same arithmetic as
// torgb_fwd_kernel for y.  Needs Ci % 16 == 0 and W % 4 == 0 (a thread's pixels share a row).
// A thread owns TS_PX = 2 consecutive pixels: their two records of a channel block are 128 contiguous bytes, but stored from
// the owning lane every store instruction would write 16-byte pieces at a 128-byte stride (with 4 pixels per thread: 256).
// Full waves pass the records through a per-wave LDS buffer and store them as slot tasks — 64 lanes write 64 CONSECUTIVE slots,
// 1 KB per instruction.  Slot k of owner lane L sits at 8 L + (k ^ (L & 7)): ds_write_b128 is served in groups of 8 contiguous
// lanes on 32 dword banks (8 different slots per group), ds_read_b128 in the 16-lane groups {0-3,12-15,20-27}, ... on 64 banks
// (MI355X_MICROARCH.md, LDS): the XOR makes both conflict free (round 3 rotated by L / 2: 2-way on every write, 31 % of the
// LDS-active cycles).
constexpr int TS_PX = 2;
__global__ __launch_bounds__(256) void torgb_fwd_sform_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ s, int s_stride, const float* __restrict__ bias,
                                                              const float* __restrict__ skip, const float* __restrict__ kern,
                                                              float* __restrict__ y, uint4* __restrict__ ys,
                                                              const float* __restrict__ ys_scale, int ys_scale_stride, SDims yd, int Ci,
                                                              int H, int W, float scale, unsigned* __restrict__ vmax) {
    __shared__ float ws[3 * kMaxCi];
    __shared__ float sc[kMaxCi];
    __shared__ float kf[16];
    __shared__ __attribute__((aligned(16))) uint4 xbuf[4][64 * TS_PX * 4];      // per wave: 64 lanes x 2 records x 4 slots
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    for (int e = threadIdx.x; e < 3 * Ci; e += 256) {
        const int c = e / Ci, ci = e % Ci;
        ws[e] = scale * w[c * Ci + ci] * s[(long)b * s_stride + ci];
    }
    for (int e = threadIdx.x; e < Ci; e += 256) sc[e] = ys_scale ? ys_scale[(long)b * ys_scale_stride + e] : 1.f;
    if (threadIdx.x < 16 && skip) kf[threadIdx.x] = kern[(3 - threadIdx.x / 4) * 4 + (3 - threadIdx.x % 4)];  // flipped
    __syncthreads();
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * TS_PX;
    const bool full_wave = __all(p < HW);        // evaluated before any lane leaves
    if (p >= HW) return;
    const float* xp = x + (long)b * Ci * HW + p;
    const int Y0 = (int)(p / W), X0 = (int)(p % W);
    float a0[TS_PX] = {0, 0}, a1[TS_PX] = {0, 0}, a2[TS_PX] = {0, 0};
    float vm = 0.f;
    // slot tasks of the record stores (full waves): task t = i*64 + lane -> owner lane t/8, its pixel (t%8)/4, slot t%4: the
    // wave's 128 records are consecutive in memory (W % 128 == 0 or not: the unit is computed per task)
    const int lane = threadIdx.x & 63;
    uint4* xb = xbuf[threadIdx.x >> 6];
    uint4* ysb = ys + sform_unit(yd, b, 0, -1, -1, 0);          // first unit of the sample's channel block 0
    int rel[2 * TS_PX * 4 / 2];
    if (full_wave) {
        const long pw = p - (long)TS_PX * lane;
#pragma unroll
        for (int i = 0; i < TS_PX * 4; ++i) {
            const int t = i * 64 + lane;
            const long po = pw + TS_PX * (t >> 3) + ((t >> 2) & 1);
            rel[i] = (int)sform_unit(yd, 0, 0, (int)(po / W), (int)(po % W), t & 3);
        }
    }
    // the 16 channel planes of block kc+1 are requested before block kc is processed: one block at a time left every block a
    // bare round trip to HBM (80 % of the wave time waiting at 3.7 TB/s)
    float2 nx[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) nx[c] = ld2_stream(xp + (long)c * HW);
    const int nkc = Ci / 16;
    for (int kc = 0; kc < nkc; ++kc) {
        unsigned hp[TS_PX][8], lp[TS_PX][8];
        float2 cx[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) cx[c] = nx[c];
        if (kc + 1 < nkc) {
#pragma unroll
            for (int c = 0; c < 16; ++c) nx[c] = ld2_stream(xp + (long)((kc + 1) * 16 + c) * HW);
        }
#pragma unroll
        for (int cp = 0; cp < 8; ++cp) {
            const int ci = kc * 16 + 2 * cp;
            const float2 v0 = cx[2 * cp];
            const float2 v1 = cx[2 * cp + 1];
            const float e0[TS_PX] = {v0.x, v0.y}, e1[TS_PX] = {v1.x, v1.y};
            const float w00 = ws[ci], w01 = ws[Ci + ci], w02 = ws[2 * Ci + ci];
            const float w10 = ws[ci + 1], w11 = ws[Ci + ci + 1], w12 = ws[2 * Ci + ci + 1];
            const float s0 = sc[ci], s1 = sc[ci + 1];
#pragma unroll
            for (int j = 0; j < TS_PX; ++j) {
                a0[j] += w00 * e0[j]; a1[j] += w01 * e0[j]; a2[j] += w02 * e0[j];
                a0[j] += w10 * e1[j]; a1[j] += w11 * e1[j]; a2[j] += w12 * e1[j];
                split_pair(e0[j] * s0, e1[j] * s1, hp[j][cp], lp[j][cp]);
                vm = fmaxf(vm, fmaxf(fabsf(e0[j] * s0), fabsf(e1[j] * s1)));
            }
        }
        if (full_wave) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = k >> 2, sl = k & 3, o = 4 * (sl & 1);
                const unsigned* src = (sl & 2) ? lp[j] : hp[j];
                xb[lane * 8 + (k ^ (lane & 7))] = make_uint4(src[o], src[o + 1], src[o + 2], src[o + 3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = i * 64 + lane, lo_ = t >> 3, k = t & 7;
                st16_stream(ysb + rel[i] + (long)kc * yd.plane, xb[lo_ * 8 + (k ^ (lo_ & 7))]);
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int j = 0; j < TS_PX; ++j) {
                uint4* rec = ys + sform_unit(yd, b, kc, Y0, X0 + j, 0);
                rec[0] = make_uint4(hp[j][0], hp[j][1], hp[j][2], hp[j][3]);
                rec[1] = make_uint4(hp[j][4], hp[j][5], hp[j][6], hp[j][7]);
                rec[2] = make_uint4(lp[j][0], lp[j][1], lp[j][2], lp[j][3]);
                rec[3] = make_uint4(lp[j][4], lp[j][5], lp[j][6], lp[j][7]);
            }
        }
    }
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f, b2 = bias ? bias[2] : 0.f;
    const int h2 = H >> 1, w2_ = W >> 1;
    // up-sampled skip: the 2 x 2 taps of both pixels are fetched from clamped positions, all 24 values in flight together, and
    // a tap outside the image gets a zero weight (a conditional load per tap was a round trip per tap: eight in a row)
    float sv[TS_PX][4][3], skw[TS_PX][4];
    if (skip) {
        const float* sp = skip + (long)b * 3 * h2 * w2_;
#pragma unroll
        for (int j = 0; j < TS_PX; ++j) {
            const int Y = Y0, X = X0 + j;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ky = (Y & 1) + 2 * t;
                const int iy = (Y + ky - 2) >> 1;
                const bool yok = Y + ky - 2 >= 0 && iy < h2;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kx = (X & 1) + 2 * u;
                    const int ix = (X + kx - 2) >> 1;
                    const bool ok = yok && X + kx - 2 >= 0 && ix < w2_;
                    const long q = (long)min(max(iy, 0), h2 - 1) * w2_ + min(max(ix, 0), w2_ - 1);
                    skw[j][2 * t + u] = ok ? kf[ky * 4 + kx] : 0.f;
                    sv[j][2 * t + u][0] = sp[q];
                    sv[j][2 * t + u][1] = sp[(long)h2 * w2_ + q];
                    sv[j][2 * t + u][2] = sp[2L * h2 * w2_ + q];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TS_PX; ++j) {
        float o0 = a0[j] + b0, o1 = a1[j] + b1, o2 = a2[j] + b2;
        if (skip) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o0 += skw[j][k] * sv[j][k][0];
                o1 += skw[j][k] * sv[j][k][1];
                o2 += skw[j][k] * sv[j][k][2];
            }
        }
        float* yp = y + (long)b * 3 * HW + p + j;
        yp[0] = o0; yp[HW] = o1; yp[2 * HW] = o2;
    }
    if (vmax) {
        if (full_wave) record_vmax(vmax, b, vm);
        else if (vm > 0.f) atomicMax(vmax + (long)b * OODGAN_VMAX_SLOTS + (threadIdx.x & (OODGAN_VMAX_SLOTS - 1)), __float_as_uint(vm));
    }
}

// second half of ToRGB when the colour sums were formed in the conv kernel's epilogue: + bias + 2x FIR-upsampled skip
__global__ __launch_bounds__(256) void rgb_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                         const float* __restrict__ skip, const float* __restrict__ kern,
                                                         float* __restrict__ y, int B, int H, int W, int nparts) {
    __shared__ float kf[16];
    if (threadIdx.x < 16 && skip) kf[threadIdx.x] = kern[(3 - threadIdx.x / 4) * 4 + (3 - threadIdx.x % 4)];  // flipped
    __syncthreads();
    const long HW = (long)H * W, total = (long)B * HW;
    const int h2 = H >> 1, w2_ = W >> 1;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int b = (int)(e / HW);
        const long p = e % HW;
        const int Y = (int)(p / W), X = (int)(p % W);
        float o0 = part[(long)b * 3 * HW + p], o1 = part[(long)b * 3 * HW + HW + p], o2 = part[(long)b * 3 * HW + 2 * HW + p];
        for (int q = 1; q < nparts; ++q) {          // partial sums of the conv's channel blocks, in the order of their index
            const float* pq = part + (long)q * B * 3 * HW + (long)b * 3 * HW + p;
            o0 += pq[0]; o1 += pq[HW]; o2 += pq[2 * HW];
        }
        if (bias) { o0 += bias[0]; o1 += bias[1]; o2 += bias[2]; }
        if (skip) {
            SkipTaps st;
            st.fetch(skip + (long)b * 3 * h2 * w2_, h2, w2_, Y, X, [&](int ky, int kx) { return kf[ky * 4 + kx]; });
            st.add(o0, o1, o2);
        }
        y[(long)b * 3 * HW + p] = o0;
        y[(long)b * 3 * HW + HW + p] = o1;
        y[(long)b * 3 * HW + 2 * HW + p] = o2;
    }
}

}  // namespace

extern "C" int oodgan_torgb_fwd(const float* x, const float* w, const float* s, int s_stride, const float* bias,
                                const float* skip, const float* kernel, float* y, int B, int Ci, int H, int W, float scale,
                                void* stream) {
    OODGAN_REQUIRE(x && w && s && y && B > 0 && Ci > 0 && H > 0 && W > 0, "torgb_fwd: bad args");
    OODGAN_REQUIRE(Ci <= kMaxCi, "torgb_fwd: Ci %d > %d", Ci, kMaxCi);
    OODGAN_REQUIRE(!skip || (kernel && (H % 2 == 0) && (W % 2 == 0)), "torgb_fwd: skip needs kernel and even H,W");
    const long HW = (long)H * W;
    if (HW <= 4096 && (HW & 3) == 0) {
        dim3 grid((unsigned)((HW / 4 + 3) / 4), B);
        hipLaunchKernelGGL(torgb_fwd_small_kernel, grid, dim3(256), 0, as_stream(stream), x, w, s, s_stride, bias, skip, kernel, y,
                           Ci, H, W, scale);
        return check_launch("torgb_fwd");
    }
    dim3 grid((unsigned)((HW + 1023) / 1024), B);
    hipLaunchKernelGGL(torgb_fwd_kernel, grid, dim3(256), 0, as_stream(stream), x, w, s, s_stride, bias, skip, kernel, y, Ci, H,
                       W, scale);
    return check_launch("torgb_fwd");
}

extern "C" int oodgan_torgb_fwd_sform(const float* x, const float* w, const float* s, int s_stride, const float* bias,
                                      const float* skip, const float* kernel, float* y, void* ys, const float* ys_scale,
                                      int ys_scale_stride, int B, int Ci, int H, int W, float scale, unsigned* vmax, void* stream) {
    OODGAN_REQUIRE(x && w && s && y && ys && B > 0 && Ci > 0 && H > 0 && W > 0, "torgb_fwd_sform: bad args");
    OODGAN_REQUIRE(Ci <= kMaxCi && (Ci % 16) == 0 && (W % 4) == 0, "torgb_fwd_sform: needs Ci %% 16 == 0, Ci <= %d, W %% 4 == 0", kMaxCi);
    OODGAN_REQUIRE(!skip || (kernel && (H % 2 == 0) && (W % 2 == 0)), "torgb_fwd_sform: skip needs kernel and even H,W");
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 256 * TS_PX - 1) / (256 * TS_PX)), B);
    hipLaunchKernelGGL(torgb_fwd_sform_kernel, grid, dim3(256), 0, as_stream(stream), x, w, s, s_stride, bias, skip, kernel, y,
                       reinterpret_cast<uint4*>(ys), ys_scale, ys_scale_stride, sform_dims(Ci, H, W), Ci, H, W, scale, vmax);
    return check_launch("torgb_fwd_sform");
}

extern "C" int oodgan_rgb_finish(const float* partial, const float* bias, const float* skip, const float* kernel, float* y, int B,
                                 int H, int W, void* stream) {
    OODGAN_REQUIRE(partial && y && B > 0 && H > 0 && W > 0, "rgb_finish: bad args");
    OODGAN_REQUIRE(!skip || (kernel && (H % 2 == 0) && (W % 2 == 0)), "rgb_finish: skip needs kernel and even H,W");
    hipLaunchKernelGGL(rgb_finish_kernel, dim3(stream_grid((long)B * H * W, 256)), dim3(256), 0, as_stream(stream), partial, bias, skip,
                       kernel, y, B, H, W, 1);
    return check_launch("rgb_finish");
}

extern "C" int oodgan_rgb_finish_parts(const float* partial, int nparts, const float* bias, const float* skip, const float* kernel, float* y,
                                       int B, int H, int W, void* stream) {
    OODGAN_REQUIRE(partial && y && nparts >= 1 && B > 0 && H > 0 && W > 0, "rgb_finish_parts: bad args");
    OODGAN_REQUIRE(!skip || (kernel && (H % 2 == 0) && (W % 2 == 0)), "rgb_finish_parts: skip needs kernel and even H,W");
    hipLaunchKernelGGL(rgb_finish_kernel, dim3(stream_grid((long)B * H * W, 256)), dim3(256), 0, as_stream(stream), partial, bias, skip,
                       kernel, y, B, H, W, nparts);
    return check_launch("rgb_finish_parts");
}
