// HBM-bound streaming kernels of the generator path: bias+noise+LeakyReLU (fwd/bwd), the merged
// activation/ToRGB backward, MSE loss + gradient, Adam, and the partial-sum reducer.
// All are float4-vectorised grid-stride loops (16 B/lane, coalesced), one (b,c) row per block-row so
// the per-channel constants are scalar.
#include "common.hpp"

using namespace oodgan;

namespace {

__device__ __forceinline__ float lrelu_s(float v, float slope, float scale) { return (v > 0.f ? v : v * slope) * scale; }

// grid: (chunks, B*C).  Each block-row handles one (b,c) plane of HW elements.
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                           const float* __restrict__ noise, const float* __restrict__ noise_w,
                                                           float* __restrict__ y, int C, long HW, int noise_batch,
                                                           float slope, float scale) {
    const int bc = blockIdx.y;
    const int b = bc / C, c = bc % C;
    const float bv = bias ? bias[c] : 0.f;
    const float nw = noise ? (noise_w ? noise_w[0] : 1.f) : 0.f;
    const float* xp = x + (long)bc * HW;
    float* yp = y + (long)bc * HW;
    const float* np = noise ? noise + (long)(noise_batch > 1 ? b : 0) * HW : nullptr;
    const long stride = (long)gridDim.x * blockDim.x;
    if ((HW & 3) == 0) {
        const long n4 = HW >> 2;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(xp)[i];
            float4 n = np ? reinterpret_cast<const float4*>(np)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x = lrelu_s(v.x + nw * n.x + bv, slope, scale);
            v.y = lrelu_s(v.y + nw * n.y + bv, slope, scale);
            v.z = lrelu_s(v.z + nw * n.z + bv, slope, scale);
            v.w = lrelu_s(v.w + nw * n.w + bv, slope, scale);
            reinterpret_cast<float4*>(yp)[i] = v;
        }
    } else {
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += stride)
            yp[i] = lrelu_s(xp[i] + (np ? nw * np[i] : 0.f) + bv, slope, scale);
    }
}

__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                           float* __restrict__ gx, long n, float slope, float scale) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(gy)[i];
        const float4 o = reinterpret_cast<const float4*>(y)[i];
        g.x *= (o.x > 0.f ? scale : slope * scale);
        g.y *= (o.y > 0.f ? scale : slope * scale);
        g.z *= (o.z > 0.f ? scale : slope * scale);
        g.w *= (o.w > 0.f ? scale : slope * scale);
        reinterpret_cast<float4*>(gx)[i] = g;
    }
    for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride)
        gx[i] = gy[i] * (y[i] > 0.f ? scale : slope * scale);
}

// gbias[c] = sum_{b,p} gx[b,c,p]; one block per channel (small C*HW in practice: tests / module API)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ gx, float* __restrict__ gbias, int B,
                                                          int C, long HW) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* p = gx + ((long)b * C + c) * HW;
        for (long i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) gbias[c] = s;
}

constexpr int kActChunk = 4096;   // elements of one (b,c) plane per block in act_bwd_fused

// grid: (nparts, B*C)
__global__ __launch_bounds__(256) void act_bwd_fused_kernel(
    const float* __restrict__ g_feat, const float* __restrict__ out, const float* __restrict__ noise, int noise_batch,
    const float* __restrict__ noise_w, const float* __restrict__ bias, const float* __restrict__ g_rgb,
    const float* __restrict__ w_rgb, const float* __restrict__ s_rgb, int s_rgb_stride, float rgb_scale,
    float* __restrict__ g_pre, float* __restrict__ part_r, float* __restrict__ part_rgb, float* __restrict__ part_max,
    const float* __restrict__ dscale, int dscale_stride, int C, long HW, int nparts) {
    __shared__ float red[4];
    const int bc = blockIdx.y, b = bc / C, c = bc % C;
    const long base = (long)bc * HW;
    const long p0 = (long)blockIdx.x * kActChunk;
    const long p1 = p0 + kActChunk < HW ? p0 + kActChunk : HW;
    const float bv = bias ? bias[c] : 0.f;
    const float nw = noise ? (noise_w ? noise_w[0] : 1.f) : 0.f;
    const float* np = noise ? noise + (long)(noise_batch > 1 ? b : 0) * HW : nullptr;
    float w0 = 0.f, w1 = 0.f, w2 = 0.f, sr = 0.f;
    const float* gr = nullptr;
    if (g_rgb) {
        w0 = w_rgb[0 * C + c] * rgb_scale;
        w1 = w_rgb[1 * C + c] * rgb_scale;
        w2 = w_rgb[2 * C + c] * rgb_scale;
        sr = s_rgb[(long)b * s_rgb_stride + c];
        gr = g_rgb + (long)b * 3 * HW;
    }
    const float inv_pos = 1.f / kSqrt2, inv_neg = 1.f / (0.2f * kSqrt2);
    float acc_r = 0.f, acc_t = 0.f, amax = 0.f;
    auto one = [&](float gf, float o, float nz, float r0, float r1, float r2) -> float {
        const float t = w0 * r0 + w1 * r1 + w2 * r2;
        const float g = gf + sr * t;
        const float gp = g * (o > 0.f ? kSqrt2 : 0.2f * kSqrt2);
        const float ycv = (o > 0.f ? o * inv_pos : o * inv_neg) - nw * nz - bv;
        acc_r += gp * ycv;
        acc_t += o * t;
        amax = fmaxf(amax, fabsf(gp));
        return gp;
    };
    if ((HW & 3) == 0) {
        for (long i = (p0 >> 2) + threadIdx.x; i < (p1 >> 2); i += 256) {
            const float4 o = reinterpret_cast<const float4*>(out + base)[i];
            float4 gf = g_feat ? reinterpret_cast<const float4*>(g_feat + base)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 nz = np ? reinterpret_cast<const float4*>(np)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
            if (gr) {
                r0 = reinterpret_cast<const float4*>(gr)[i];
                r1 = reinterpret_cast<const float4*>(gr + HW)[i];
                r2 = reinterpret_cast<const float4*>(gr + 2 * HW)[i];
            }
            float4 gp;
            gp.x = one(gf.x, o.x, nz.x, r0.x, r1.x, r2.x);
            gp.y = one(gf.y, o.y, nz.y, r0.y, r1.y, r2.y);
            gp.z = one(gf.z, o.z, nz.z, r0.z, r1.z, r2.z);
            gp.w = one(gf.w, o.w, nz.w, r0.w, r1.w, r2.w);
            reinterpret_cast<float4*>(g_pre + base)[i] = gp;
        }
    } else {
        for (long i = p0 + threadIdx.x; i < p1; i += 256) {
            const float o = out[base + i];
            g_pre[base + i] = one(g_feat ? g_feat[base + i] : 0.f, o, np ? np[i] : 0.f, gr ? gr[i] : 0.f,
                                  gr ? gr[HW + i] : 0.f, gr ? gr[2 * HW + i] : 0.f);
        }
    }
    const float sr_ = block_sum_256(acc_r, red);
    if (threadIdx.x == 0 && part_r) part_r[(long)bc * nparts + blockIdx.x] = sr_;
    if (part_rgb) {
        const float st_ = block_sum_256(acc_t, red);
        if (threadIdx.x == 0) part_rgb[(long)bc * nparts + blockIdx.x] = st_;
    }
    if (part_max) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0)
            part_max[(long)bc * nparts + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) *
                                                        (dscale ? fabsf(dscale[(long)b * dscale_stride + c]) : 1.f);
    }
}

template <bool CLEAR>
__global__ __launch_bounds__(256) void absmax_scale_kernel(float* __restrict__ part, long n, float* __restrict__ out2) {
    __shared__ float red[4];
    float m = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        m = fmaxf(m, fabsf(part[i]));
        if (CLEAR) part[i] = 0.f;           // the slots are ready for the next measurement: no fill launch per use
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        int e = 0;
        if (m > 0.f && isfinite(m)) e = 9 - (int)floorf(log2f(m));
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        out2[0] = ldexpf(1.f, -e);
        out2[1] = ldexpf(1.f, e);
    }
}

// out[row] (+)= sum_j part[row, j]; one wave per row
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ part, float* __restrict__ out, long rows,
                                                           int nparts, int accumulate) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float s = row_sum(part + row * nparts, nparts, lane);        // common.hpp: the order oodgan_reduce_batch uses
    if (lane == 0) out[row] = accumulate ? out[row] + s : s;
}

// out[b*out_stride + c] (+)= sum_j part[(b*C + c), j]: the reduction lands directly in a column block of a wider matrix
// (the style-gradient accumulator), so no separate copy / add pass is needed; one wave per (b,c)
__global__ __launch_bounds__(256) void reduce_parts_cols_kernel(const float* __restrict__ part, float* __restrict__ out, int B, int C,
                                                                int nparts, int out_stride, int accumulate) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * C) return;
    const int lane = threadIdx.x & 63;
    const float* p = part + row * nparts;
    float s = 0.f;
    for (int j = lane; j < nparts; j += 64) s += p[j];
    s = wave_sum(s);
    if (lane == 0) {
        float* o = out + (row / C) * out_stride + (row % C);
        *o = accumulate ? *o + s : s;
    }
}

constexpr int kMseChunk = 16384;

// grid: (nparts, B)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ img, const float* __restrict__ target,
                                                  float* __restrict__ gimg, float* __restrict__ part, long CHW, int nparts,
                                                  float gscale) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long base = (long)b * CHW;
    const long p0 = (long)blockIdx.x * kMseChunk;
    const long p1 = p0 + kMseChunk < CHW ? p0 + kMseChunk : CHW;
    float acc = 0.f;
    if ((CHW & 3) == 0) {
        for (long i = (p0 >> 2) + threadIdx.x; i < (p1 >> 2); i += 256) {
            const float4 a = reinterpret_cast<const float4*>(img + base)[i];
            const float4 t = reinterpret_cast<const float4*>(target + base)[i];
            float4 d = make_float4(a.x - t.x, a.y - t.y, a.z - t.z, a.w - t.w);
            acc += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
            if (gimg) reinterpret_cast<float4*>(gimg + base)[i] = make_float4(d.x * gscale, d.y * gscale, d.z * gscale, d.w * gscale);
        }
    } else {
        for (long i = p0 + threadIdx.x; i < p1; i += 256) {
            const float d = img[base + i] - target[base + i];
            acc += d * d;
            if (gimg) gimg[base + i] = d * gscale;
        }
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) part[(long)b * nparts + blockIdx.x] = acc;
}

// row_dev != NULL: the losses go to row min(row_dev[0], nrows - 1) of a (nrows, B) table — the W+ loop's loss table indexed by its
// device step counter, so that a recorded / replayed step (oodgan_plan_run, hipGraph) writes a new row each time
__global__ __launch_bounds__(64) void mse_finish_kernel(const float* __restrict__ part, float* __restrict__ loss, int nparts,
                                                        float inv_n, const int* __restrict__ row_dev, int nrows) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int j = lane; j < nparts; j += 64) s += part[(long)b * nparts + j];
    s = wave_sum(s);
    const long row = row_dev ? (long)min(max(row_dev[0], 0), nrows - 1) * gridDim.x : 0;
    if (lane == 0) loss[row + b] = s * inv_n;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float beta1, float beta2, float eps,
                                                   float step_size, float inv_bc2_sqrt) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        // torch.optim.Adam (single-tensor path): lerp for m, addcmul for v
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        w[i] = w[i] - step_size * (mi / denom);
    }
}

// Adam with the step index kept on the device (so that a whole W+ step can be replayed from a hipGraph):
// t_dev[0] is incremented by a one-thread kernel, then every thread derives the bias corrections from it.
__global__ void counter_inc_kernel(int* __restrict__ t) { t[0] += 1; }

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, float lr, float beta1, float beta2,
                                                       float eps, const int* __restrict__ t_dev) {
    const int t = t_dev[0];
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        w[i] = w[i] - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
    }
}

}  // namespace

extern "C" int oodgan_adam_step_dev(float* w, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                                    float eps, int* t_dev, void* stream) {
    OODGAN_REQUIRE(w && g && m && v && t_dev && n > 0, "adam_dev: bad args");
    hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, as_stream(stream), t_dev);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), w, g, m, v, n, lr, beta1,
                       beta2, eps, t_dev);
    return check_launch("adam_dev");
}

extern "C" int oodgan_bias_act_fwd(const float* x, const float* bias, const float* noise, const float* noise_w, float* y,
                                   int B, int C, long HW, int noise_batch, float slope, float scale, void* stream) {
    OODGAN_REQUIRE(x && y && B > 0 && C > 0 && HW > 0, "bias_act_fwd: bad args");
    OODGAN_REQUIRE(noise == nullptr || noise_batch == 1 || noise_batch == B, "bias_act_fwd: noise_batch %d", noise_batch);
    OODGAN_REQUIRE((long)B * C < 65536L * 16, "bias_act_fwd: B*C too large");
    long per_row = (HW + 3) / 4;
    int gx = (int)((per_row + 255) / 256);
    if (gx > 64) gx = 64;
    dim3 grid(gx, B * C);
    hipLaunchKernelGGL(bias_act_fwd_kernel, grid, dim3(256), 0, as_stream(stream), x, bias, noise, noise_w, y, C, HW,
                       noise_batch, slope, scale);
    return check_launch("bias_act_fwd");
}

extern "C" int oodgan_bias_act_bwd(const float* gy, const float* y, float* gx, float* gbias, int B, int C, long HW,
                                   float slope, float scale, void* stream) {
    OODGAN_REQUIRE(gy && y && gx && B > 0 && C > 0 && HW > 0, "bias_act_bwd: bad args");
    const long n = (long)B * C * HW;
    hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(stream_grid((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), gy, y, gx,
                       n, slope, scale);
    int rc = check_launch("bias_act_bwd");
    if (rc != OODGAN_OK || !gbias) return rc;
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, as_stream(stream), gx, gbias, B, C, HW);
    return check_launch("bias_act_bwd/gbias");
}

namespace {
// feature_modulation (reference src/ops/StyleGAN/model.py:588-610), clss = 1: mode 0 SFT y = x*(1+c0) + c1,
// 1 ADD y = x + c1, 2 FUSE y = x + c1*sigmoid(c0)
__global__ __launch_bounds__(256) void feature_modulation_kernel(const float* __restrict__ x, const float* __restrict__ c0,
                                                                 const float* __restrict__ c1, float* __restrict__ y, long n, int mode) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i], b = c1[i];
        float o;
        if (mode == 0) o = v * (1.f + c0[i]) + b;
        else if (mode == 1) o = v + b;
        else o = v + b * (1.f / (1.f + expf(-c0[i])));
        y[i] = o;
    }
}
}  // namespace

extern "C" int oodgan_feature_modulation(const float* x, const float* c0, const float* c1, float* y, long n, int mode, void* stream) {
    OODGAN_REQUIRE(x && c1 && y && n > 0 && mode >= 0 && mode <= 2 && (mode == 1 || c0), "feature_modulation: bad args");
    hipLaunchKernelGGL(feature_modulation_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), x, c0, c1, y, n, mode);
    return check_launch("feature_modulation");
}

extern "C" int oodgan_act_bwd_nparts(long HW) { return (int)((HW + kActChunk - 1) / kActChunk); }

extern "C" int oodgan_act_bwd_fused_max(const float* g_feat, const float* out, const float* noise, int noise_batch,
                                        const float* noise_w, const float* bias, const float* g_rgb, const float* w_rgb,
                                        const float* s_rgb, int s_rgb_stride, float rgb_scale, float* g_pre, float* part_r,
                                        float* part_rgb, float* part_max, const float* dscale, int dscale_stride, int B, int C,
                                        long HW, void* stream) {
    OODGAN_REQUIRE(out && g_pre && B > 0 && C > 0 && HW > 0, "act_bwd_fused: bad args");
    OODGAN_REQUIRE(g_feat || g_rgb, "act_bwd_fused: no incoming gradient");
    OODGAN_REQUIRE(!g_rgb || (w_rgb && s_rgb), "act_bwd_fused: rgb branch needs w_rgb and s_rgb");
    OODGAN_REQUIRE(noise == nullptr || noise_batch == 1 || noise_batch == B, "act_bwd_fused: noise_batch");
    OODGAN_REQUIRE((long)B * C <= 65535, "act_bwd_fused: B*C too large");
    const int nparts = oodgan_act_bwd_nparts(HW);
    dim3 grid(nparts, B * C);
    hipLaunchKernelGGL(act_bwd_fused_kernel, grid, dim3(256), 0, as_stream(stream), g_feat, out, noise, noise_batch, noise_w,
                       bias, g_rgb, w_rgb, s_rgb, s_rgb_stride, rgb_scale, g_pre, part_r, g_rgb ? part_rgb : nullptr, part_max,
                       dscale, dscale_stride, C, HW, nparts);
    return check_launch("act_bwd_fused");
}

extern "C" int oodgan_act_bwd_fused(const float* g_feat, const float* out, const float* noise, int noise_batch,
                                    const float* noise_w, const float* bias, const float* g_rgb, const float* w_rgb,
                                    const float* s_rgb, int s_rgb_stride, float rgb_scale, float* g_pre, float* part_r,
                                    float* part_rgb, int B, int C, long HW, void* stream) {
    return oodgan_act_bwd_fused_max(g_feat, out, noise, noise_batch, noise_w, bias, g_rgb, w_rgb, s_rgb, s_rgb_stride, rgb_scale,
                                    g_pre, part_r, part_rgb, nullptr, nullptr, 0, B, C, HW, stream);
}

extern "C" int oodgan_absmax_scale(const float* part, long n, float* out2, void* stream) {
    OODGAN_REQUIRE(part && out2 && n > 0, "absmax_scale: bad args");
    hipLaunchKernelGGL(absmax_scale_kernel<false>, dim3(1), dim3(256), 0, as_stream(stream), const_cast<float*>(part), n, out2);
    return check_launch("absmax_scale");
}

extern "C" int oodgan_absmax_scale_clear(float* part, long n, float* out2, void* stream) {
    OODGAN_REQUIRE(part && out2 && n > 0, "absmax_scale_clear: bad args");
    hipLaunchKernelGGL(absmax_scale_kernel<true>, dim3(1), dim3(256), 0, as_stream(stream), part, n, out2);
    return check_launch("absmax_scale_clear");
}

extern "C" int oodgan_reduce_parts(const float* part, float* out, long rows, int nparts, int accumulate, void* stream) {
    OODGAN_REQUIRE(part && out && rows > 0 && nparts > 0, "reduce_parts: bad args");
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, as_stream(stream), part, out, rows,
                       nparts, accumulate);
    return check_launch("reduce_parts");
}

extern "C" int oodgan_reduce_parts_cols(const float* part, float* out, int B, int C, int nparts, int out_stride, int accumulate,
                                        void* stream) {
    OODGAN_REQUIRE(part && out && B > 0 && C > 0 && nparts > 0 && out_stride >= C, "reduce_parts_cols: bad args");
    hipLaunchKernelGGL(reduce_parts_cols_kernel, dim3((unsigned)(((long)B * C + 3) / 4)), dim3(256), 0, as_stream(stream), part, out, B,
                       C, nparts, out_stride, accumulate);
    return check_launch("reduce_parts_cols");
}

extern "C" int oodgan_mse_nparts(long CHW) { return (int)((CHW + kMseChunk - 1) / kMseChunk); }

extern "C" int oodgan_mse_fwd_bwd(const float* img, const float* target, float* gimg, float* part, float* loss, int B,
                                  long CHW, float grad_mul, void* stream) {
    OODGAN_REQUIRE(img && target && part && loss && B > 0 && CHW > 0, "mse: bad args");
    const int nparts = oodgan_mse_nparts(CHW);
    hipLaunchKernelGGL(mse_kernel, dim3(nparts, B), dim3(256), 0, as_stream(stream), img, target, gimg, part, CHW, nparts,
                       grad_mul * 2.0f / (float)CHW);
    int rc = check_launch("mse");
    if (rc != OODGAN_OK) return rc;
    hipLaunchKernelGGL(mse_finish_kernel, dim3(B), dim3(64), 0, as_stream(stream), part, loss, nparts, 1.0f / (float)CHW,
                       (const int*)nullptr, 1);
    return check_launch("mse_finish");
}

extern "C" int oodgan_mse_fwd_bwd_row(const float* img, const float* target, float* gimg, float* part, float* loss_table,
                                      const int* row_dev, int nrows, int B, long CHW, float grad_mul, void* stream) {
    OODGAN_REQUIRE(img && target && part && loss_table && row_dev && nrows > 0 && B > 0 && CHW > 0, "mse_row: bad args");
    const int nparts = oodgan_mse_nparts(CHW);
    hipLaunchKernelGGL(mse_kernel, dim3(nparts, B), dim3(256), 0, as_stream(stream), img, target, gimg, part, CHW, nparts,
                       grad_mul * 2.0f / (float)CHW);
    int rc = check_launch("mse");
    if (rc != OODGAN_OK) return rc;
    hipLaunchKernelGGL(mse_finish_kernel, dim3(B), dim3(64), 0, as_stream(stream), part, loss_table, nparts, 1.0f / (float)CHW, row_dev,
                       nrows);
    return check_launch("mse_finish");
}

extern "C" int oodgan_adam_step(float* w, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                                float eps, int t, void* stream) {
    OODGAN_REQUIRE(w && g && m && v && n > 0 && t >= 1, "adam: bad args");
    const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), w, g, m, v, n, beta1, beta2,
                       eps, step_size, inv_bc2_sqrt);
    return check_launch("adam");
}
