// SAMM / SAIM (spatial alignment + invertibility mask) feature-decomposition ops and the final
// mask-compose + blend.  reference: src/ops/SAMM/helpers.py:62-179, OOD_faceGAN_e4e_arch.py:315-347.
// Index math (nearest / bilinear / bicubic source indices, linspace grid) restates ATen's float
// formulas so that masks match the reference bit-for-bit where they are pure indexing.
#include "common.hpp"

using namespace oodgan;

namespace {

// centred second moment of four values, every operation rounded on its own: the three kernels that produce InstanceNorm statistics
// (alone, with the residual sum, with the aligned pair) must agree to the bit, whatever the compiler would contract in each of them
__device__ __forceinline__ float sqsum4(float x, float y, float z, float w, float mean) {
    const float a = __fsub_rn(x, mean), b = __fsub_rn(y, mean), c = __fsub_rn(z, mean), d = __fsub_rn(w, mean);
    return __fadd_rn(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)), __fadd_rn(__fmul_rn(c, c), __fmul_rn(d, d)));
}
__device__ __forceinline__ float sqacc(float q, float x, float mean) {
    const float a = __fsub_rn(x, mean);
    return __fadd_rn(q, __fmul_rn(a, a));
}

// ---------------------------------------------------------------- InstanceNorm statistics
// one block per (b,c) plane; two passes (mean, then centred second moment) — the plane (<=256 KB)
// is served from L2 on the second pass.
__global__ __launch_bounds__(256) void instnorm_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, long HW,
                                                             float eps) {
    __shared__ float red[4];
    __shared__ float mean_s;
    const float* p = x + (long)blockIdx.x * HW;
    float s = 0.f;
    if ((HW & 3) == 0) {
        // four float4 per thread in flight per trip (clamped index, masked by a select): one per trip was a round trip per trip
        const long n4 = HW >> 2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (long i = threadIdx.x; i < n4; i += 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4*>(p)[min(i + 256 * u, n4 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float t = i + 256 * u < n4 ? (v[u].x + v[u].y) + (v[u].z + v[u].w) : 0.f;
                if (u == 0) s0 += t; else if (u == 1) s1 += t; else if (u == 2) s2 += t; else s3 += t;
            }
        }
        s = (s0 + s1) + (s2 + s3);
    } else {
        for (long i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) mean_s = s / (float)HW;
    __syncthreads();
    const float mean = mean_s;
    float q = 0.f;
    if ((HW & 3) == 0) {
        const long n4 = HW >> 2;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
        for (long i = threadIdx.x; i < n4; i += 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4*>(p)[min(i + 256 * u, n4 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float t = i + 256 * u < n4 ? sqsum4(v[u].x, v[u].y, v[u].z, v[u].w, mean) : 0.f;
                if (u == 0) q0 += t; else if (u == 1) q1 += t; else if (u == 2) q2 += t; else q3 += t;
            }
        }
        q = (q0 + q1) + (q2 + q3);
    } else {
        for (long i = threadIdx.x; i < HW; i += 256) q = sqacc(q, p[i], mean);
    }
    q = block_sum_256(q, red);
    if (threadIdx.x == 0) {
        stats[2 * (long)blockIdx.x] = mean;
        stats[2 * (long)blockIdx.x + 1] = rsqrtf(q / (float)HW + eps);
    }
}

__global__ void instnorm_coeffs_kernel(const float* __restrict__ stats, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ sc, float* __restrict__ sh, int B,
                                       int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int c = i % C;
    const float mean = stats[2 * i], rstd = stats[2 * i + 1];
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    sc[i] = rstd * g;
    sh[i] = bt - mean * rstd * g;
}

// grid (chunks, B*C)
__global__ __launch_bounds__(256) void affine_apply_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                           const float* __restrict__ sh, const float* __restrict__ res,
                                                           float* __restrict__ y, long HW) {
    const long base = (long)blockIdx.y * HW;
    const float a = sc[blockIdx.y], b = sh[blockIdx.y];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < HW; i += (long)gridDim.x * 256)
        y[base + i] = x[base + i] * a + b + (res ? res[base + i] : 0.f);
}

// y = x*sc + sh + res (the InstanceNorm + shortcut sum that ends a bottleneck) AND the InstanceNorm statistics of y for the
// bottleneck that follows, in one pass: one block per (b,c) plane, the sums in the order of instnorm_stats_kernel (bit-identical
// statistics); the second moment re-reads the thread's own y values (L2).  Saves the statistics pass over the 2C-channel tensor.
__global__ __launch_bounds__(256) void affine_apply_stats_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                                 const float* __restrict__ sh, const float* __restrict__ res,
                                                                 float* __restrict__ y, float* __restrict__ stats, long HW, float eps) {
    __shared__ float red[4];
    __shared__ float mean_s;
    const long base = (long)blockIdx.x * HW;
    const float a = sc[blockIdx.x], b = sh[blockIdx.x];
    const float* xp = x + base;
    const float* rp = res ? res + base : nullptr;
    float* yp = y + base;
    float s = 0.f;
    const bool vec = (HW & 3) == 0;
    if (vec) {
        const long n4 = HW >> 2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (long i = threadIdx.x; i < n4; i += 1024) {
            float4 v[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = reinterpret_cast<const float4*>(xp)[min(i + 256 * u, n4 - 1)];
                r[u] = rp ? reinterpret_cast<const float4*>(rp)[min(i + 256 * u, n4 - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 o;
                o.x = v[u].x * a + b + r[u].x; o.y = v[u].y * a + b + r[u].y; o.z = v[u].z * a + b + r[u].z; o.w = v[u].w * a + b + r[u].w;
                const bool live = i + 256 * u < n4;
                if (live) reinterpret_cast<float4*>(yp)[i + 256 * u] = o;
                const float t = live ? (o.x + o.y) + (o.z + o.w) : 0.f;
                if (u == 0) s0 += t; else if (u == 1) s1 += t; else if (u == 2) s2 += t; else s3 += t;
            }
        }
        s = (s0 + s1) + (s2 + s3);
    } else {
        for (long i = threadIdx.x; i < HW; i += 256) {
            const float o = xp[i] * a + b + (rp ? rp[i] : 0.f);
            yp[i] = o;
            s += o;
        }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) mean_s = s / (float)HW;
    __syncthreads();
    const float mean = mean_s;
    float q = 0.f;
    if (vec) {
        const long n4 = HW >> 2;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
        for (long i = threadIdx.x; i < n4; i += 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const float4*>(yp)[min(i + 256 * u, n4 - 1)];     // this thread's own stores
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float t = i + 256 * u < n4 ? sqsum4(v[u].x, v[u].y, v[u].z, v[u].w, mean) : 0.f;
                if (u == 0) q0 += t; else if (u == 1) q1 += t; else if (u == 2) q2 += t; else q3 += t;
            }
        }
        q = (q0 + q1) + (q2 + q3);
    } else {
        for (long i = threadIdx.x; i < HW; i += 256) q = sqacc(q, yp[i], mean);
    }
    q = block_sum_256(q, red);
    if (threadIdx.x == 0) {
        stats[2 * (long)blockIdx.x] = mean;
        stats[2 * (long)blockIdx.x + 1] = rsqrtf(q / (float)HW + eps);
    }
}

// IN(gen) - IN(enc) and IN(enc) of one element, every operation rounded on its own (as the reference's separate instance_norm / sub ops
// do; a contracted multiply-subtract would differ between the two kernels below)
__device__ __forceinline__ float align_e(float e, float me, float re) { return __fmul_rn(__fsub_rn(e, me), re); }
__device__ __forceinline__ float align_d(float g, float mg, float rg, float en) { return __fsub_rn(__fmul_rn(__fsub_rn(g, mg), rg), en); }

// grid (chunks, B*C): out[b, c] = IN(gen)-IN(enc), out[b, C+c] = IN(enc)
__global__ __launch_bounds__(256) void align_input_kernel(const float* __restrict__ gen, const float* __restrict__ enc,
                                                          const float* __restrict__ sg, const float* __restrict__ se,
                                                          float* __restrict__ out, int C, long HW, int diff) {
    const int bc = blockIdx.y, b = bc / C, c = bc % C;
    const float mg = sg[2 * bc], rg = sg[2 * bc + 1], me = se[2 * bc], re = se[2 * bc + 1];
    const float* gp = gen + (long)bc * HW;
    const float* ep = enc + (long)bc * HW;
    float* o0 = out + ((long)b * 2 * C + c) * HW;
    float* o1 = out + ((long)b * 2 * C + C + c) * HW;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        const float e = align_e(ep[i], me, re);
        o0[i] = align_d(gp[i], mg, rg, diff ? e : 0.f);        // diff_fAndg = False (helpers.py:98-101): IN(gen) itself (x - 0 is exact)
        o1[i] = e;
    }
}

// align_input AND the InstanceNorm statistics of its two output planes (the norm in front of AlignNet's first conv), one block per
// (b, c): the sums in instnorm_stats_kernel's order over the values just stored (bit-identical statistics), second moment from the
// thread's own stores.  Saves the statistics pass over the 2C-channel tensor.
__global__ __launch_bounds__(256) void align_input_stats_kernel(const float* __restrict__ gen, const float* __restrict__ enc,
                                                                const float* __restrict__ sg, const float* __restrict__ se,
                                                                float* __restrict__ out, float* __restrict__ stats, int C, long HW, float eps, int diff) {
    __shared__ float red[4];
    __shared__ float mean_s[2];
    const int bc = blockIdx.x, b = bc / C, c = bc % C;
    const float mg = sg[2 * bc], rg = sg[2 * bc + 1], me = se[2 * bc], re = se[2 * bc + 1];
    const float* gp = gen + (long)bc * HW;
    const float* ep = enc + (long)bc * HW;
    float* o0 = out + ((long)b * 2 * C + c) * HW;
    float* o1 = out + ((long)b * 2 * C + C + c) * HW;
    float sa = 0.f, sb = 0.f;
    const bool vec = (HW & 3) == 0;
    if (vec) {
        const long n4 = HW >> 2;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        for (long i = threadIdx.x; i < n4; i += 1024) {
            float4 g4[4], e4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g4[u] = reinterpret_cast<const float4*>(gp)[min(i + 256 * u, n4 - 1)];
                e4[u] = reinterpret_cast<const float4*>(ep)[min(i + 256 * u, n4 - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 e_, d_;
                e_.x = align_e(e4[u].x, me, re); e_.y = align_e(e4[u].y, me, re); e_.z = align_e(e4[u].z, me, re); e_.w = align_e(e4[u].w, me, re);
                d_.x = align_d(g4[u].x, mg, rg, diff ? e_.x : 0.f); d_.y = align_d(g4[u].y, mg, rg, diff ? e_.y : 0.f);
                d_.z = align_d(g4[u].z, mg, rg, diff ? e_.z : 0.f); d_.w = align_d(g4[u].w, mg, rg, diff ? e_.w : 0.f);
                const bool live = i + 256 * u < n4;
                if (live) {
                    reinterpret_cast<float4*>(o0)[i + 256 * u] = d_;
                    reinterpret_cast<float4*>(o1)[i + 256 * u] = e_;
                }
                const float ta = live ? (d_.x + d_.y) + (d_.z + d_.w) : 0.f, tb = live ? (e_.x + e_.y) + (e_.z + e_.w) : 0.f;
                if (u == 0) { a0 += ta; b0 += tb; } else if (u == 1) { a1 += ta; b1 += tb; } else if (u == 2) { a2 += ta; b2 += tb; } else { a3 += ta; b3 += tb; }
            }
        }
        sa = (a0 + a1) + (a2 + a3);
        sb = (b0 + b1) + (b2 + b3);
    } else {
        for (long i = threadIdx.x; i < HW; i += 256) {
            const float e_ = align_e(ep[i], me, re), d_ = align_d(gp[i], mg, rg, diff ? e_ : 0.f);
            o0[i] = d_; o1[i] = e_;
            sa += d_; sb += e_;
        }
    }
    sa = block_sum_256(sa, red);
    sb = block_sum_256(sb, red);
    if (threadIdx.x == 0) { mean_s[0] = sa / (float)HW; mean_s[1] = sb / (float)HW; }
    __syncthreads();
    const float ma = mean_s[0], mb = mean_s[1];
    float qa = 0.f, qb = 0.f;
    if (vec) {
        const long n4 = HW >> 2;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        for (long i = threadIdx.x; i < n4; i += 1024) {
            float4 d4[4], e4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                       // this thread's own stores
                d4[u] = reinterpret_cast<const float4*>(o0)[min(i + 256 * u, n4 - 1)];
                e4[u] = reinterpret_cast<const float4*>(o1)[min(i + 256 * u, n4 - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool live = i + 256 * u < n4;
                const float ta = live ? sqsum4(d4[u].x, d4[u].y, d4[u].z, d4[u].w, ma) : 0.f, tb = live ? sqsum4(e4[u].x, e4[u].y, e4[u].z, e4[u].w, mb) : 0.f;
                if (u == 0) { a0 += ta; b0 += tb; } else if (u == 1) { a1 += ta; b1 += tb; } else if (u == 2) { a2 += ta; b2 += tb; } else { a3 += ta; b3 += tb; }
            }
        }
        qa = (a0 + a1) + (a2 + a3);
        qb = (b0 + b1) + (b2 + b3);
    } else {
        for (long i = threadIdx.x; i < HW; i += 256) {
            qa = sqacc(qa, o0[i], ma);
            qb = sqacc(qb, o1[i], mb);
        }
    }
    qa = block_sum_256(qa, red);
    qb = block_sum_256(qb, red);
    if (threadIdx.x == 0) {
        float* s0 = stats + 2 * ((long)b * 2 * C + c);
        float* s1 = stats + 2 * ((long)b * 2 * C + C + c);
        s0[0] = ma; s0[1] = rsqrtf(qa / (float)HW + eps);
        s1[0] = mb; s1[1] = rsqrtf(qb / (float)HW + eps);
    }
}

// ---------------------------------------------------------------- dense 1x1 conv
// block: 256 pixels x 16 output channels; weights for the 16 rows staged in LDS per K chunk of 64.
constexpr int C1_MT = 16, C1_KC = 64;
__global__ __launch_bounds__(256) void conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int K, int M,
                                                      long HW) {
    __shared__ float ws[C1_MT * C1_KC];
    const int b = blockIdx.z, m0 = blockIdx.y * C1_MT;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const float* xp = x + (long)b * K * HW + p;
    float acc[C1_MT];
#pragma unroll
    for (int j = 0; j < C1_MT; ++j) acc[j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += C1_KC) {
        __syncthreads();
        for (int e = threadIdx.x; e < C1_MT * C1_KC; e += 256) {
            const int j = e / C1_KC, kk = e % C1_KC;
            ws[e] = (m0 + j < M && k0 + kk < K) ? w[(long)(m0 + j) * K + k0 + kk] : 0.f;
        }
        __syncthreads();
        if (p < HW) {
            const int kn = (K - k0) < C1_KC ? (K - k0) : C1_KC;
            // eight input planes in flight per trip (clamped channel, masked by a select); one plane per trip was a round trip per
            // input channel.  Same order of the sums as before.
            for (int kk0 = 0; kk0 < kn; kk0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = xp[(long)(k0 + min(kk0 + u, kn - 1)) * HW];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float vu = kk0 + u < kn ? v[u] : 0.f;
                    const int kk = min(kk0 + u, C1_KC - 1);
#pragma unroll
                    for (int j = 0; j < C1_MT; ++j) acc[j] += ws[j * C1_KC + kk] * vu;
                }
            }
        }
    }
    if (p < HW) {
#pragma unroll
        for (int j = 0; j < C1_MT; ++j)
            if (m0 + j < M) y[((long)b * M + m0 + j) * HW + p] = acc[j] + (bias ? bias[m0 + j] : 0.f);
    }
}

// ---- wide form (round 4) — block: 256 pixels x MT = 64 output channels, thread = one pixel; the weights of a K chunk of 32 staged in LDS
// as [k][channel] so that FOUR channels come with one (broadcast) ds_read_b128, and x is read M/64 instead of M/16 times (the narrow
// form above pays one ds_read_b32 per multiply-add: 36-47 TFLOP/s whatever the shape).  Taken where the grid still has >= 1024 blocks;
// the sums run in the order of k as in the narrow form (same results).
constexpr int C1W_KC = 32;
template <int MT>
__global__ __launch_bounds__(256) void conv1x1_wide_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int K, int M,
                                                      long HW) {
    constexpr int WP = MT + 4;                                     // pitch: 16-byte aligned rows, writes spread over the banks
    __shared__ __attribute__((aligned(16))) float ws[C1W_KC * WP];
    const int b = blockIdx.z, m0 = blockIdx.y * MT;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const long pc = p < HW ? p : HW - 1;
    const float* xp = x + (long)b * K * HW + pc;
    float acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += C1W_KC) {
        const int kn = (K - k0) < C1W_KC ? (K - k0) : C1W_KC;
        __syncthreads();
        for (int e = threadIdx.x; e < MT * C1W_KC; e += 256) {     // global reads along k, transposed into [k][channel]
            const int j = e / C1W_KC, kk = e % C1W_KC;
            ws[kk * WP + j] = (m0 + j < M && kk < kn) ? w[(long)(m0 + j) * K + k0 + kk] : 0.f;
        }
        __syncthreads();
        for (int kk0 = 0; kk0 < kn; kk0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float t = xp[(long)(k0 + min(kk0 + u, kn - 1)) * HW];
                v[u] = kk0 + u < kn ? t : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* wr = ws + min(kk0 + u, C1W_KC - 1) * WP;
#pragma unroll
                for (int j4 = 0; j4 < MT / 4; ++j4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wr + 4 * j4);
                    acc[4 * j4] += w4.x * v[u]; acc[4 * j4 + 1] += w4.y * v[u]; acc[4 * j4 + 2] += w4.z * v[u]; acc[4 * j4 + 3] += w4.w * v[u];
                }
            }
        }
    }
    if (p < HW) {
#pragma unroll
        for (int j = 0; j < MT; ++j)
            if (m0 + j < M) y[((long)b * M + m0 + j) * HW + p] = acc[j] + (bias ? bias[m0 + j] : 0.f);
    }
}

// ---------------------------------------------------------------- squeeze-excitation gate of bottleneck_IR_SE
// gate[b,c] = sigmoid(W2 relu(W1 mean_b))  (SEModule.forward, src/ops/e4e/encoders/helpers.py:60-76: AdaptiveAvgPool2d(1) -> fc1 ->
// ReLU -> fc2 -> Sigmoid; the product with the input is the caller's oodgan_affine_apply).  One workgroup per sample.  As two
// oodgan_conv1x1 launches on a 1x1 "image" plus a ReLU and a sigmoid launch the gate cost 70 us per block (one active thread per
// workgroup walking K): 24 blocks = 10 % of the B = 1 forward.
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ stats, const float* __restrict__ w1,
                                                      const float* __restrict__ w2, float* __restrict__ gate, int C, int Cr) {
    __shared__ float mean[1024], hid[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) mean[c] = stats[((long)b * C + c) * 2];
    __syncthreads();
    const int sub = tid & 7;
    for (int j0 = 0; j0 < Cr; j0 += 32) {
        const int j = j0 + (tid >> 3);
        float acc = 0.f;
        if (j < Cr)
            for (int c = sub; c < C; c += 8) acc += w1[(long)j * C + c] * mean[c];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (sub == 0 && j < Cr) hid[j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float acc = 0.f;
        for (int j = 0; j < Cr; ++j) acc += w2[(long)c * Cr + j] * hid[j];
        gate[(long)b * C + c] = 1.f / (1.f + expf(-acc));
    }
}

// ---------------------------------------------------------------- small direct 3x3 (K,M <= 8)
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ in_sc, const float* __restrict__ in_sh,
                                                            const float* __restrict__ slope, float* __restrict__ y, int K, int M,
                                                            int H, int W) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int py = (int)(p / W), px = (int)(p % W);
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.f;
    for (int k = 0; k < K; ++k) {
        const float sc = in_sc ? in_sc[b * K + k] : 1.f, sh = in_sh ? in_sh[b * K + k] : 0.f;
        const float* xp = x + ((long)b * K + k) * HW;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = py + ky - 1, ix = px + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float v = xp[(long)iy * W + ix] * sc + sh;
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    if (m < M) acc[m] += w[((m * K + k) * 3 + ky) * 3 + kx] * v;
            }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
        if (m < M) {
            float v = acc[m];
            if (slope) v = v > 0.f ? v : slope[m] * v;
            y[((long)b * M + m) * HW + p] = v;
        }
}

// ---------------------------------------------------------------- 3x3 conv from MANY input channels to a FEW outputs (M <= 4)
// The head conv of AlignNet's second bottleneck: 2C -> 3 channels, C = 512 ... 128 at 32² ... 256² (reference
// src/ops/SAMM/helpers.py:58-60 -> bottleneck_IR(2C, 3), src/ops/e4e/encoders/helpers.py:439-444), with InstanceNorm's affine
// as in_sc / in_sh (shift on in-bounds samples only = norm followed by zero padding).  It reads K*H*W values once and
// writes 3*H*W: HBM-bound streaming work, exact fp32 — on the matrix kernels it was a 32-channel M tile with 3 live rows
// and, at 32² / 64², four to sixteen workgroups walking 64 K-chunks one after the other (327 us per launch at B = 1).
// Here the K range is split over gridDim.y so that every level fills the chip; partial sums are combined in a fixed
// order by conv3x3_fewout_finish_kernel (deterministic, no atomics), which also applies the PReLU.
constexpr int FO_TH = 8, FO_TW = 32, FO_KC = 8, FO_P = FO_TW + 4;      // 8 x 32 pixel tile, 8 channels per LDS stage
__global__ __launch_bounds__(256) void conv3x3_fewout_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ in_sc, const float* __restrict__ in_sh,
                                                             float* __restrict__ part, int K, int M, int H, int W, int kslice,
                                                             int tiles_x) {
    __shared__ float tile[FO_KC][FO_TH + 2][FO_P];
    __shared__ float wl[FO_KC][9][4];
    const int b = blockIdx.z, ks = blockIdx.y, KS = gridDim.y;
    const int r0 = (blockIdx.x / tiles_x) * FO_TH, c0 = (blockIdx.x % tiles_x) * FO_TW;
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const long HW = (long)H * W;
    const int k_begin = ks * kslice, k_end = min(K, k_begin + kslice);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = k_begin; k0 < k_end; k0 += FO_KC) {
        __syncthreads();
        for (int e = tid; e < FO_KC * (FO_TH + 2) * (FO_TW + 2); e += 256) {
            const int c = e / ((FO_TH + 2) * (FO_TW + 2)), r = (e / (FO_TW + 2)) % (FO_TH + 2), q = e % (FO_TW + 2);
            const int k = k0 + c, iy = r0 + r - 1, ix = c0 + q - 1;
            float v = 0.f;
            if (k < k_end && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = x[((long)b * K + k) * HW + (long)iy * W + ix];
                if (in_sc) v *= in_sc[(long)b * K + k];
                if (in_sh) v += in_sh[(long)b * K + k];
            }
            tile[c][r][q] = v;
        }
        for (int e = tid; e < FO_KC * 9 * 4; e += 256) {
            const int c = e / 36, t = (e / 4) % 9, m = e & 3, k = k0 + c;
            wl[c][t][m] = (k < k_end && m < M) ? w[((long)m * K + k) * 9 + t] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < FO_KC; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = tile[c][ty + t / 3][tx + t % 3];
                const float4 wv = *reinterpret_cast<const float4*>(&wl[c][t][0]);
                acc[0] += wv.x * v; acc[1] += wv.y * v; acc[2] += wv.z * v; acc[3] += wv.w * v;
            }
    }
    const int py = r0 + ty, px = c0 + tx;
    if (py < H && px < W)
        for (int m = 0; m < M; ++m) part[(((long)b * KS + ks) * M + m) * HW + (long)py * W + px] = acc[m];
}

// Second form (round 4), K % 8 == 0: the element -> (channel, row, column) decomposition of the tile load is done ONCE per thread (it
// cost more instructions per stage than the 288 multiply-adds), the weights are read through the scalar cache from a (K, 9, 4)
// transposed copy (uniform address: s_load_dwordx4 instead of an LDS broadcast read per tap), and — X11 — the 1x1 shortcut conv of
// the same bottleneck (bottleneck_IR.shortcut_layer[0], 2C -> 3 on the RAW input, e4e helpers.py:431-437) is accumulated from the
// same pass: one read of the 2C-channel tensor for both convs.
template <bool X11>
__global__ __launch_bounds__(256) void conv3x3_fewout2_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                              const float* __restrict__ w11t, const float* __restrict__ in_sc,
                                                              const float* __restrict__ in_sh, float* __restrict__ part,
                                                              float* __restrict__ part2, int K, int M, int M2, int H, int W, int kslice,
                                                              int tiles_x) {
    constexpr int NE = FO_KC * (FO_TH + 2) * (FO_TW + 2), NI = (NE + 255) / 256;       // 2720 elements, 11 per thread
    __shared__ float tile[FO_KC * (FO_TH + 2) * FO_P];
    const int b = blockIdx.z, ks = blockIdx.y, KS = gridDim.y;
    const int r0 = (blockIdx.x / tiles_x) * FO_TH, c0 = (blockIdx.x % tiles_x) * FO_TW;
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    const long HW = (long)H * W;
    const int k_begin = ks * kslice, k_end = min(K, k_begin + kslice);
    int goff[NI], loff[NI], cch[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = tid + 256 * i;
        const int c = e / ((FO_TH + 2) * (FO_TW + 2)), rem = e % ((FO_TH + 2) * (FO_TW + 2));
        const int r = rem / (FO_TW + 2), q = rem % (FO_TW + 2);
        const int iy = r0 + r - 1, ix = c0 + q - 1;
        const bool ok = e < NE && iy >= 0 && iy < H && ix >= 0 && ix < W;
        goff[i] = ok ? (int)((long)c * HW + (long)iy * W + ix) : -1;        // FO_KC planes: fits an int (HW <= 2^27)
        loff[i] = e < NE ? (c * (FO_TH + 2) + r) * FO_P + q : -1;
        cch[i] = c < FO_KC ? c : FO_KC - 1;
    }
    const int py = r0 + ty, px = c0 + tx;
    const bool pok = py < H && px < W;
    const long poff = (long)min(py, H - 1) * W + min(px, W - 1);
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, acc2[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xb = x + (long)b * K * HW;
    // software pipeline: the loads of stage t+1 are in flight while stage t is computed (one stage at a time this kernel was a chain
    // of K/8 exposed memory latencies per workgroup: 280 us per launch at B = 8 for 35 us of LDS / FMA work)
    float v[NI], sc_[NI], sh_[NI], raw[FO_KC];
    auto request = [&](int k0) {
        const float* xk = xb + (long)k0 * HW;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            v[i] = goff[i] >= 0 ? xk[goff[i]] : 0.f;
            sc_[i] = in_sc ? in_sc[(long)b * K + k0 + cch[i]] : 1.f;
            sh_[i] = in_sh ? in_sh[(long)b * K + k0 + cch[i]] : 0.f;
        }
        if (X11) {
#pragma unroll
            for (int c = 0; c < FO_KC; ++c) raw[c] = xk[(long)c * HW + poff];
        }
    };
    if (k_begin < k_end) request(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += FO_KC) {
        __syncthreads();                 // the previous stage's reads of the tile are done
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (loff[i] >= 0) tile[loff[i]] = goff[i] >= 0 ? v[i] * sc_[i] + sh_[i] : 0.f;
        float rawc[FO_KC];
        if (X11) {
#pragma unroll
            for (int c = 0; c < FO_KC; ++c) rawc[c] = raw[c];
        }
        __syncthreads();
        if (k0 + FO_KC < k_end) request(k0 + FO_KC);
#pragma unroll
        for (int c = 0; c < FO_KC; ++c) {
            const float* wk = wt + (long)(k0 + c) * 36;         // uniform: scalar loads
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float tv = tile[(c * (FO_TH + 2) + ty + t / 3) * FO_P + tx + t % 3];
                acc[0] += wk[4 * t] * tv; acc[1] += wk[4 * t + 1] * tv; acc[2] += wk[4 * t + 2] * tv; acc[3] += wk[4 * t + 3] * tv;
            }
            if (X11) {
                const float* w1 = w11t + (long)(k0 + c) * 4;
                acc2[0] += w1[0] * rawc[c]; acc2[1] += w1[1] * rawc[c]; acc2[2] += w1[2] * rawc[c]; acc2[3] += w1[3] * rawc[c];
            }
        }
    }
    if (pok) {
        for (int m = 0; m < M; ++m) part[(((long)b * KS + ks) * M + m) * HW + (long)py * W + px] = acc[m];
        if (X11)
            for (int m = 0; m < M2; ++m) part2[(((long)b * KS + ks) * M2 + m) * HW + (long)py * W + px] = acc2[m];
    }
}

// Third form (round 4), M == 3, K % 8 == 0, W % 4 == 0: the second form moved one dword per lane and instruction everywhere — 41 loads, 72
// LDS reads and 320 multiply-adds per thread and stage, all three near a CU's rate limits at once (1.0-1.2 TB/s of input).  Here a
// thread owns FOUR horizontally adjacent pixels and a wave owns TWO of the stage's eight channels:
//   * tile load: the wave fetches its own two channels (340 elements each, six 64-lane trips) — the channel of a trip is uniform, so
//     InstanceNorm's scale / shift are scalar loads, not 22 vector loads per thread;
//   * a 3 x 6 window per channel is three ds_read_b128 + three ds_read_b64 for 36 taps x 3 outputs (12 LDS reads per stage, not 72);
//   * the 1x1 conv's raw inputs are one float4 per channel;
//   * the four waves' partial sums (24 per thread) are added through LDS at the end.
template <bool X11>
__global__ __launch_bounds__(256) void conv3x3_fewout3_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                              const float* __restrict__ w11t, const float* __restrict__ in_sc,
                                                              const float* __restrict__ in_sh, float* __restrict__ part,
                                                              float* __restrict__ part2, int K, int H, int W, int kslice, int tiles_x) {
    constexpr int TR = FO_TH + 2, TC = FO_TW + 2, NT = 6;            // 10 x 34 elements per channel, six trips of 64 lanes
    __shared__ __attribute__((aligned(16))) float tile[FO_KC * TR * FO_P];
    __shared__ float red[4 * 24 * 64];
    const int b = blockIdx.z, ks = blockIdx.y, KS = gridDim.y;
    const int r0 = (blockIdx.x / tiles_x) * FO_TH, c0 = (blockIdx.x % tiles_x) * FO_TW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qy = lane >> 3, qx = lane & 7;
    const long HW = (long)H * W;
    const int k_begin = ks * kslice, k_end = min(K, k_begin + kslice);
    int goff[NT], loff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int e = j * 64 + lane;
        const int r = e / TC, q = e % TC;
        const int iy = r0 + r - 1, ix = c0 + q - 1;
        const bool in = e < TR * TC;
        goff[j] = (in && iy >= 0 && iy < H && ix >= 0 && ix < W) ? iy * W + ix : -1;
        loff[j] = in ? r * FO_P + q : -1;
    }
    const int py = r0 + qy, px = c0 + 4 * qx;
    const bool pok = py < H && px < W;               // W % 4 == 0: the whole quad is in or out
    const long poff = (long)min(py, H - 1) * W + min(px, W - 4);
    float acc[4][3], acc2[4][3];
#pragma unroll
    for (int p_ = 0; p_ < 4; ++p_)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[p_][m] = acc2[p_][m] = 0.f;
    const float* xb = x + (long)b * K * HW;
    float v[2][NT];
    float4 raw[2];
    auto request = [&](int k0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* xk = xb + (long)(k0 + 2 * wave + c) * HW;
#pragma unroll
            for (int j = 0; j < NT; ++j) v[c][j] = goff[j] >= 0 ? xk[goff[j]] : 0.f;
            if (X11) raw[c] = *reinterpret_cast<const float4*>(xk + poff);
        }
    };
    if (k_begin < k_end) request(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += FO_KC) {
        __syncthreads();                 // the previous stage's reads of the tile are done
        float4 rawc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = k0 + 2 * wave + c;                       // uniform
            const float sc = in_sc ? in_sc[(long)b * K + k] : 1.f, sh = in_sh ? in_sh[(long)b * K + k] : 0.f;
            float* tl = tile + (2 * wave + c) * (TR * FO_P);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (loff[j] >= 0) tl[loff[j]] = goff[j] >= 0 ? v[c][j] * sc + sh : 0.f;
            rawc[c] = raw[c];
        }
        __syncthreads();
        if (k0 + FO_KC < k_end) request(k0 + FO_KC);
        // every wave walks all eight channels of the stage for ITS pixels?  No: a wave's threads cover the whole 8 x 32 tile (64 quads),
        // and the wave adds the contribution of its two channels only; the other six come from the other waves (summed at the end)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = 2 * wave + c;
            const float* wk = wt + (long)(k0 + ch) * 36;           // uniform: scalar loads
            const float* tl = tile + ch * (TR * FO_P) + qy * FO_P + 4 * qx;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float4 a4 = *reinterpret_cast<const float4*>(tl + r * FO_P);
                const float2 b2 = *reinterpret_cast<const float2*>(tl + r * FO_P + 4);
                const float w6[6] = {a4.x, a4.y, a4.z, a4.w, b2.x, b2.y};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float w0 = wk[4 * (3 * r + kx)], w1 = wk[4 * (3 * r + kx) + 1], w2 = wk[4 * (3 * r + kx) + 2];
#pragma unroll
                    for (int p_ = 0; p_ < 4; ++p_) {
                        const float tv = w6[p_ + kx];
                        acc[p_][0] += w0 * tv; acc[p_][1] += w1 * tv; acc[p_][2] += w2 * tv;
                    }
                }
            }
            if (X11) {
                const float* w1p = w11t + (long)(k0 + ch) * 4;
                const float u0 = w1p[0], u1 = w1p[1], u2 = w1p[2];
                const float rq[4] = {rawc[c].x, rawc[c].y, rawc[c].z, rawc[c].w};
#pragma unroll
                for (int p_ = 0; p_ < 4; ++p_) { acc2[p_][0] += u0 * rq[p_]; acc2[p_][1] += u1 * rq[p_]; acc2[p_][2] += u2 * rq[p_]; }
            }
        }
    }
    // the four waves' partial sums -> one (fixed order: wave 0 + 1 + 2 + 3)
#pragma unroll
    for (int p_ = 0; p_ < 4; ++p_)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            red[(wave * 24 + m * 4 + p_) * 64 + lane] = acc[p_][m];
            if (X11) red[(wave * 24 + 12 + m * 4 + p_) * 64 + lane] = acc2[p_][m];
        }
    __syncthreads();
    // thread (wave = output index group, lane = quad): outputs m*4 + p for the 3x3 conv (0..11) and 12 + m*4 + p for the 1x1 conv
    const int nout = X11 ? 6 : 3;
    for (int o = wave; o < nout; o += 4) {                         // o = m (3x3) or 3 + m (1x1): one float4 of four pixels
        float4 sum;
        float* sp = reinterpret_cast<float*>(&sum);
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) {
            const int idx = (o < 3 ? o * 4 : 12 + (o - 3) * 4) + p_;
            sp[p_] = ((red[(0 * 24 + idx) * 64 + lane] + red[(1 * 24 + idx) * 64 + lane]) + red[(2 * 24 + idx) * 64 + lane]) + red[(3 * 24 + idx) * 64 + lane];
        }
        if (pok) {
            float* dst = (o < 3 ? part : part2) + (((long)b * KS + ks) * 3 + (o < 3 ? o : o - 3)) * HW + (long)py * W + px;
            *reinterpret_cast<float4*>(dst) = sum;
        }
    }
}

__global__ __launch_bounds__(256) void conv3x3_fewout_finish_kernel(const float* __restrict__ part, const float* __restrict__ slope,
                                                                    float* __restrict__ y, int KS, int M, long HW, long total) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long p = e % HW;
        const int m = (int)((e / HW) % M);
        const long b = e / (HW * M);
        float v = 0.f;
        for (int ks = 0; ks < KS; ++ks) v += part[((b * KS + ks) * M + m) * HW + p];
        if (slope) v = v > 0.f ? v : slope[m] * v;
        y[e] = v;
    }
}

__global__ __launch_bounds__(256) void align_head_kernel(const float* __restrict__ x, float* __restrict__ y, long HW, long total,
                                                         float scale) {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)((e / HW) % 3);
        const float v = x[e];
        y[e] = c < 2 ? tanhf(v) * scale : 1.f / (1.f + expf(-v));
    }
}

// ---------------------------------------------------------------- resampling helpers (ATen float formulas)
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// bicubic, align_corners=True, A=-0.75, border-clamped taps (F.interpolate(mode='bicubic', align_corners=True))
__device__ float bicubic_ac(const float* __restrict__ src, int Hs, int Ws, int oy, int ox, int Ho, int Wo) {
    const float A = -0.75f;
    const float sy = Ho > 1 ? (float)(Hs - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(Ws - 1) / (float)(Wo - 1) : 0.f;
    const float ry = sy * oy, rx = sx * ox;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    const float ty = ry - iy, tx = rx - ix;
    float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
    float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int yy = iy - 1 + i;
        yy = yy < 0 ? 0 : (yy > Hs - 1 ? Hs - 1 : yy);
        float row = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int xx = ix - 1 + j;
            xx = xx < 0 ? 0 : (xx > Ws - 1 ? Ws - 1 : xx);
            row += wx[j] * src[(long)yy * Ws + xx];
        }
        acc += wy[i] * row;
    }
    return acc;
}

// bilinear, align_corners=False (F.interpolate(mode='bilinear'))
__device__ float bilinear_nc(const float* __restrict__ src, int Hs, int Ws, int oy, int ox, int Ho, int Wo) {
    const float sy = (float)Hs / (float)Ho, sx = (float)Ws / (float)Wo;
    float ry = sy * (oy + 0.5f) - 0.5f, rx = sx * (ox + 0.5f) - 0.5f;
    ry = ry < 0.f ? 0.f : ry;
    rx = rx < 0.f ? 0.f : rx;
    const int y0 = (int)ry, x0 = (int)rx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = ry - y0, lx = rx - x0, hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * src[(long)y0 * Ws + x0] + lx * src[(long)y0 * Ws + x1]) +
           ly * (hx * src[(long)y1 * Ws + x0] + lx * src[(long)y1 * Ws + x1]);
}

__device__ __forceinline__ float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// SPM_Warp.add / upsample_add
__global__ __launch_bounds__(256) void field_compose_kernel(const float* __restrict__ acc, const float* __restrict__ cur,
                                                            const float* __restrict__ prev, float* __restrict__ out, int B, int H,
                                                            int W, int Hp, int Wp, float scale, int mode) {
    const long HW = (long)H * W;
    const long total = (long)B * HW;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int b = (int)(e / HW);
        const long p = e % HW;
        const float* c = cur + (long)b * 3 * HW;
        float* o = out + (long)b * 3 * HW;
        float dx, dy, xa;   // xa: the "x" argument of new_PRM (coarser / accumulated alpha)
        if (mode == 0) {
            const float* a = acc + (long)b * 3 * HW;
            dx = clipf(a[p] + c[p], -scale, scale);
            dy = clipf(a[HW + p] + c[HW + p], -scale, scale);
            xa = a[2 * HW + p];
        } else {
            dx = c[p];
            dy = c[HW + p];
            const float* pa = prev + ((long)b * 3 + 2) * Hp * Wp;
            xa = (Hp == H && Wp == W) ? pa[p] : bicubic_ac(pa, Hp, Wp, (int)(p / W), (int)(p % W), H, W);
        }
        const float ya = c[2 * HW + p];
        o[p] = dx;
        o[HW + p] = dy;
        o[2 * HW + p] = clipf(ya * xa + xa * (1.f - xa), 0.f, 1.f);
    }
}

// torch.linspace(-1, 1, n)[i] in float (symmetric evaluation of ATen's kernel)
__device__ __forceinline__ float linspace_pm1(int i, int n) {
    if (n == 1) return -1.f;
    const float step = 2.f / (float)(n - 1);
    return i < n / 2 ? -1.f + step * i : 1.f - step * (n - 1 - i);
}

// grid (chunks of 256 pixels, chunks of WB_CPB channels, B); each thread one pixel and WB_CPB channels (coalesced per plane).
// The sampling weights depend on the pixel only.  (One thread per pixel looping over ALL channels left the 32² level with four
// workgroups walking 512 planes one dependent gather at a time: 335 us per launch at B = 1, 87 % of the wave time waiting.)
constexpr int WB_CPB = 8;
__global__ __launch_bounds__(256) void warp_blend_kernel(const float* __restrict__ target, const float* __restrict__ field,
                                                         float* __restrict__ y, int C, int H, int W) {
    const int b = blockIdx.z;
    const long HW = (long)H * W;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int py = (int)(p / W), px = (int)(p % W);
    const float* f = field + (long)b * 3 * HW;
    const float gx = linspace_pm1(px, W) + f[p];
    const float gy = linspace_pm1(py, H) + f[HW + p];
    const float alpha = f[2 * HW + p];
    // grid_sampler_unnormalize, align_corners=False
    const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;   // nw, ne, sw, se
    const bool v00 = vy0 && vx0, v01 = vy0 && vx1, v10 = vy1 && vx0, v11 = vy1 && vx1;
    const long o00 = (long)y0 * W + x0, o01 = (long)y0 * W + x1, o10 = (long)y1 * W + x0, o11 = (long)y1 * W + x1;
    const int cb = blockIdx.y * WB_CPB;
    const float* tb = target + ((long)b * C + cb) * HW;
    float* yb = y + ((long)b * C + cb) * HW;
    float t00[WB_CPB], t01[WB_CPB], t10[WB_CPB], t11[WB_CPB], tv[WB_CPB];
#pragma unroll
    for (int c = 0; c < WB_CPB; ++c) {          // all gathers of the chunk are issued before the first use
        const bool ok = cb + c < C;
        const float* t = tb + (long)c * HW;
        t00[c] = (ok && v00) ? t[o00] : 0.f;
        t01[c] = (ok && v01) ? t[o01] : 0.f;
        t10[c] = (ok && v10) ? t[o10] : 0.f;
        t11[c] = (ok && v11) ? t[o11] : 0.f;
        tv[c] = ok ? t[p] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < WB_CPB; ++c) {
        if (cb + c >= C) break;
        float v = 0.f;                           // the reference's accumulation order (nw, ne, sw, se)
        if (v00) v += t00[c] * w00;
        if (v01) v += t01[c] * w01;
        if (v10) v += t10[c] * w10;
        if (v11) v += t11[c] * w11;
        yb[(long)c * HW + p] = v * alpha + tv[c] * (1.f - alpha);
    }
}

struct MaskArgs {
    const float* fields[4];
    int sizes[4];
    int nfields;
};

// grid (chunks, B)
__global__ __launch_bounds__(256) void mask_blend_kernel(const MaskArgs m, const float* __restrict__ x, const float* __restrict__ gen,
                                                         float* __restrict__ alpha_out, float* __restrict__ out, int S) {
    const int b = blockIdx.y;
    const long SS = (long)S * S;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < SS; p += (long)gridDim.x * 256) {
        const int oy = (int)(p / S), ox = (int)(p % S);
        float a = 0.f;
        for (int k = 0; k < m.nfields; ++k) {
            const int s = m.sizes[k];
            const float* src = m.fields[k] + ((long)b * 3 + 2) * s * s;
            const float ak = (s == S) ? src[p] : bilinear_nc(src, s, s, oy, ox, S, S);
            a = (k == 0) ? ak : ak * a + a * (1.f - a);
        }
        a = clipf(a, 0.f, 1.f);
        if (alpha_out) alpha_out[(long)b * SS + p] = a;
        if (out) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const long q = ((long)b * 3 + c) * SS + p;
                out[q] = a * x[q] + gen[q] * (1.f - a);
            }
        }
    }
}

__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int Hin,
                                                             int Win, int Hout, int Wout, int out_pitch, int out_xoff) {
    const long total = (long)planes * Hout * Wout;
    const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % Wout), oy = (int)((e / Wout) % Hout);
        const long pl = e / ((long)Wout * Hout);
        int iy = (int)floorf(oy * sy), ix = (int)floorf(ox * sx);
        iy = iy < Hin - 1 ? iy : Hin - 1;
        ix = ix < Win - 1 ? ix : Win - 1;
        y[(pl * Hout + oy) * (long)out_pitch + out_xoff + ox] = x[(pl * Hin + iy) * (long)Win + ix];
    }
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int Hin,
                                                              int Win, int Hout, int Wout) {
    const long total = (long)planes * Hout * Wout;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % Wout), oy = (int)((e / Wout) % Hout);
        const long pl = e / ((long)Wout * Hout);
        y[e] = bilinear_nc(x + pl * Hin * Win, Hin, Win, oy, ox, Hout, Wout);
    }
}

// y = bicubic(x -> Hout x Wout, align_corners=True) (+ add): the FPN step of the e4e encoder
// (reference src/ops/e4e/encoders/helpers.py:504-521 `_upsample_add`)
__global__ __launch_bounds__(256) void resize_bicubic_ac_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                                float* __restrict__ y, int planes, int Hin, int Win, int Hout,
                                                                int Wout) {
    const long total = (long)planes * Hout * Wout;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % Wout), oy = (int)((e / Wout) % Hout);
        const long pl = e / ((long)Wout * Hout);
        const float v = bicubic_ac(x + pl * Hin * Win, Hin, Win, oy, ox, Hout, Wout);
        y[e] = add ? v + add[e] : v;
    }
}

// y = AdaptiveAvgPool2d((Hout, Wout))(x): window [floor(o*In/Out), ceil((o+1)*In/Out)) per axis.  ReStyle's `face_pool`
// (reference src/archs/OOD_faceGAN_restyle_arch.py:89, 1024 -> 256, 4x4 windows: one thread per output) and the 3x3 pooled
// descriptors of the FeatureStyle encoder (src/ops/FeatureStyle/feature_style_encoder.py:42, windows up to 43x43: one wave
// per output, lanes stride the window, DPP-free shuffle reduction).
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int Hin, int Win,
                                                      int Hout, int Wout) {
    const long total = (long)planes * Hout * Wout;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % Wout), oy = (int)((e / Wout) % Hout);
        const int y0 = (int)((long)oy * Hin / Hout), y1 = (int)(((long)(oy + 1) * Hin + Hout - 1) / Hout);
        const int x0 = (int)((long)ox * Win / Wout), x1 = (int)(((long)(ox + 1) * Win + Wout - 1) / Wout);
        const float* src = x + (e / ((long)Wout * Hout)) * Hin * Win;
        float acc = 0.f;
        for (int i = y0; i < y1; ++i)
            for (int j = x0; j < x1; ++j) acc += src[(long)i * Win + j];
        y[e] = acc / (float)((y1 - y0) * (x1 - x0));
    }
}

__global__ __launch_bounds__(256) void avgpool_wave_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int Hin,
                                                           int Win, int Hout, int Wout) {
    const long total = (long)planes * Hout * Wout;
    const int lane = threadIdx.x & 63;
    for (long e = blockIdx.x * 4L + (threadIdx.x >> 6); e < total; e += (long)gridDim.x * 4) {
        const int ox = (int)(e % Wout), oy = (int)((e / Wout) % Hout);
        const int y0 = (int)((long)oy * Hin / Hout), y1 = (int)(((long)(oy + 1) * Hin + Hout - 1) / Hout);
        const int x0 = (int)((long)ox * Win / Wout), x1 = (int)(((long)(ox + 1) * Win + Wout - 1) / Wout);
        const float* src = x + (e / ((long)Wout * Hout)) * Hin * Win;
        const int ww = x1 - x0, n = (y1 - y0) * ww;
        float acc = 0.f;
        for (int t = lane; t < n; t += 64) acc += src[(long)(y0 + t / ww) * Win + x0 + t % ww];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) y[e] = acc / (float)n;
    }
}

}  // namespace

extern "C" int oodgan_avgpool(const float* x, float* y, int planes, int Hin, int Win, int Hout, int Wout, void* stream) {
    OODGAN_REQUIRE(x && y && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Hout <= Hin && Wout <= Win, "avgpool: bad args");
    const long total = (long)planes * Hout * Wout;
    const long window = (long)((Hin + Hout - 1) / Hout + 1) * ((Win + Wout - 1) / Wout + 1);
    if (window >= 128)
        hipLaunchKernelGGL(avgpool_wave_kernel, dim3(stream_grid(total, 4)), dim3(256), 0, as_stream(stream), x, y, planes, Hin, Win, Hout,
                           Wout);
    else
        hipLaunchKernelGGL(avgpool_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), x, y, planes, Hin, Win, Hout,
                           Wout);
    return check_launch("avgpool");
}

extern "C" int oodgan_resize_bicubic_ac(const float* x, const float* add, float* y, int planes, int Hin, int Win, int Hout, int Wout,
                                        void* stream) {
    OODGAN_REQUIRE(x && y && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "resize_bicubic_ac: bad args");
    hipLaunchKernelGGL(resize_bicubic_ac_kernel, dim3(stream_grid((long)planes * Hout * Wout, 256)), dim3(256), 0,
                       as_stream(stream), x, add, y, planes, Hin, Win, Hout, Wout);
    return check_launch("resize_bicubic_ac");
}

extern "C" int oodgan_instnorm_stats(const float* x, float* stats, int B, int C, long HW, float eps, void* stream) {
    OODGAN_REQUIRE(x && stats && B > 0 && C > 0 && HW > 0, "instnorm_stats: bad args");
    hipLaunchKernelGGL(instnorm_stats_kernel, dim3(B * C), dim3(256), 0, as_stream(stream), x, stats, HW, eps);
    return check_launch("instnorm_stats");
}

extern "C" int oodgan_instnorm_coeffs(const float* stats, const float* gamma, const float* beta, float* sc, float* sh, int B,
                                      int C, void* stream) {
    OODGAN_REQUIRE(stats && sc && sh && B > 0 && C > 0, "instnorm_coeffs: bad args");
    hipLaunchKernelGGL(instnorm_coeffs_kernel, dim3((B * C + 255) / 256), dim3(256), 0, as_stream(stream), stats, gamma, beta, sc,
                       sh, B, C);
    return check_launch("instnorm_coeffs");
}

extern "C" int oodgan_affine_apply(const float* x, const float* sc, const float* sh, const float* res, float* y, int B, int C,
                                   long HW, void* stream) {
    OODGAN_REQUIRE(x && sc && sh && y && B > 0 && C > 0 && HW > 0 && (long)B * C <= 65535, "affine_apply: bad args");
    int gx = (int)((HW + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(affine_apply_kernel, dim3(gx, B * C), dim3(256), 0, as_stream(stream), x, sc, sh, res, y, HW);
    return check_launch("affine_apply");
}

extern "C" int oodgan_align_input(const float* gen, const float* enc, const float* st_gen, const float* st_enc, float* out, int diff, int B,
                                  int C, long HW, void* stream) {
    OODGAN_REQUIRE(gen && enc && st_gen && st_enc && out && B > 0 && C > 0 && HW > 0 && (long)B * C <= 65535,
                   "align_input: bad args");
    int gx = (int)((HW + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(align_input_kernel, dim3(gx, B * C), dim3(256), 0, as_stream(stream), gen, enc, st_gen, st_enc, out, C, HW, diff);
    return check_launch("align_input");
}

extern "C" int oodgan_align_input_stats(const float* gen, const float* enc, const float* st_gen, const float* st_enc, float* out, float* stats, int diff,
                                        int B, int C, long HW, float eps, void* stream) {
    OODGAN_REQUIRE(gen && enc && st_gen && st_enc && out && stats && B > 0 && C > 0 && HW > 0, "align_input_stats: bad args");
    hipLaunchKernelGGL(align_input_stats_kernel, dim3(B * C), dim3(256), 0, as_stream(stream), gen, enc, st_gen, st_enc, out, stats, C, HW, eps, diff);
    return check_launch("align_input_stats");
}

extern "C" int oodgan_conv1x1(const float* x, const float* w, const float* bias, float* y, int B, int K, int M, long HW,
                              void* stream) {
    OODGAN_REQUIRE(x && w && y && B > 0 && K > 0 && M > 0 && HW > 0, "conv1x1: bad args");
    // 64 channels per block (x read M/64 times, [k][channel] weights in LDS) when that still gives the chip four blocks per CU: 91 / 196 /
    // 155 us against 102 / 247 / 193 for the 16-channel form at 128->512 @64², 64->128 @256², 128->256 @128² (batch 8); on the small grids
    // (32² maps, batch 1) the narrow form stays ahead (53 vs 74-88 us)
    if (M > 16 && ((HW + 255) / 256) * ((M + 63) / 64) * B >= 1024) {
        dim3 grid((unsigned)((HW + 255) / 256), (M + 63) / 64, B);
        hipLaunchKernelGGL(conv1x1_wide_kernel<64>, grid, dim3(256), 0, as_stream(stream), x, w, bias, y, K, M, HW);
    } else {
        dim3 grid((unsigned)((HW + 255) / 256), (M + C1_MT - 1) / C1_MT, B);
        hipLaunchKernelGGL(conv1x1_kernel, grid, dim3(256), 0, as_stream(stream), x, w, bias, y, K, M, HW);
    }
    return check_launch("conv1x1");
}

extern "C" int oodgan_se_gate(const float* stats, const float* w1, const float* w2, float* gate, int B, int C, int Cr, void* stream) {
    OODGAN_REQUIRE(stats && w1 && w2 && gate && B > 0 && C > 0 && C <= 1024 && Cr > 0 && Cr <= 64, "se_gate: bad args (C <= 1024, C/r <= 64)");
    hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(256), 0, as_stream(stream), stats, w1, w2, gate, C, Cr);
    return check_launch("se_gate");
}

extern "C" int oodgan_conv3x3_small(const float* x, const float* w, const float* in_sc, const float* in_sh, const float* slope,
                                    float* y, int B, int K, int M, int H, int W, void* stream) {
    OODGAN_REQUIRE(x && w && y && B > 0 && K > 0 && K <= 8 && M > 0 && M <= 8 && H > 0 && W > 0, "conv3x3_small: bad args");
    dim3 grid((unsigned)(((long)H * W + 255) / 256), B);
    hipLaunchKernelGGL(conv3x3_small_kernel, grid, dim3(256), 0, as_stream(stream), x, w, in_sc, in_sh, slope, y, K, M, H, W);
    return check_launch("conv3x3_small");
}

// K-split of oodgan_conv3x3_fewout for this shape: the caller allocates part (B, ksplit, M, H, W)
extern "C" int oodgan_conv3x3_fewout_ksplit(int B, int K, int H, int W) {
    if (B <= 0 || K <= 0 || H <= 0 || W <= 0) return 0;
    const long tiles = (long)((H + FO_TH - 1) / FO_TH) * ((W + FO_TW - 1) / FO_TW) * B;
    long ks = 1024 / tiles;                       // aim at ~4 workgroups per CU
    if (ks > K / 16) ks = K / 16;
    if (ks < 1) ks = 1;
    const int kslice = (int)(((K + ks - 1) / ks + FO_KC - 1) / FO_KC * FO_KC);
    return (K + kslice - 1) / kslice;
}

extern "C" int oodgan_conv3x3_fewout(const float* x, const float* w, const float* in_sc, const float* in_sh, const float* slope,
                                     float* part, float* y, int B, int K, int M, int H, int W, void* stream) {
    OODGAN_REQUIRE(x && w && part && y && B > 0 && K > 0 && M > 0 && M <= 4 && H > 0 && W > 0, "conv3x3_fewout: bad args");
    const int KS = oodgan_conv3x3_fewout_ksplit(B, K, H, W);
    const int kslice = ((K + KS - 1) / KS + FO_KC - 1) / FO_KC * FO_KC;
    const int tiles_x = (W + FO_TW - 1) / FO_TW, tiles_y = (H + FO_TH - 1) / FO_TH;
    OODGAN_REQUIRE(B <= 65535 && KS <= 65535 && (long)(KS - 1) * kslice < K, "conv3x3_fewout: split");
    hipLaunchKernelGGL(conv3x3_fewout_kernel, dim3((unsigned)(tiles_x * tiles_y), KS, B), dim3(256), 0, as_stream(stream), x, w, in_sc,
                       in_sh, part, K, M, H, W, kslice, tiles_x);
    const long HW = (long)H * W, total = (long)B * M * HW;
    hipLaunchKernelGGL(conv3x3_fewout_finish_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), part, slope, y, KS,
                       M, HW, total);
    return check_launch("conv3x3_fewout");
}

extern "C" int oodgan_affine_apply_stats(const float* x, const float* sc, const float* sh, const float* res, float* y, float* stats, int B,
                                         int C, long HW, float eps, void* stream) {
    OODGAN_REQUIRE(x && sc && sh && y && stats && B > 0 && C > 0 && HW > 0, "affine_apply_stats: bad args");
    hipLaunchKernelGGL(affine_apply_stats_kernel, dim3(B * C), dim3(256), 0, as_stream(stream), x, sc, sh, res, y, stats, HW, eps);
    return check_launch("affine_apply_stats");
}

// wt: (K, 9, 4) transposed copy of the (M <= 4, K, 3, 3) weight (zero-filled), w11t: (K, 4) of the (M2 <= 4, K) 1x1 weight or NULL;
// part / part2: (B, ksplit, M | M2, H, W); y2 (B, M2, H, W) <- conv1x1(x) (raw input, no activation)
extern "C" int oodgan_conv3x3_fewout2(const float* x, const float* wt, const float* w11t, const float* in_sc, const float* in_sh,
                                      const float* slope, float* part, float* part2, float* y, float* y2, int B, int K, int M, int M2,
                                      int H, int W, void* stream) {
    OODGAN_REQUIRE(x && wt && part && y && B > 0 && K > 0 && K % FO_KC == 0 && M > 0 && M <= 4 && H > 0 && W > 0 && (long)H * W <= (1L << 27),
                   "conv3x3_fewout2: bad args (K %% 8 == 0, M <= 4)");
    OODGAN_REQUIRE(!w11t || (part2 && y2 && M2 > 0 && M2 <= 4), "conv3x3_fewout2: the 1x1 branch needs part2, y2 and M2 <= 4");
    const int KS = oodgan_conv3x3_fewout_ksplit(B, K, H, W);
    const int kslice = ((K + KS - 1) / KS + FO_KC - 1) / FO_KC * FO_KC;
    const int tiles_x = (W + FO_TW - 1) / FO_TW, tiles_y = (H + FO_TH - 1) / FO_TH;
    OODGAN_REQUIRE(B <= 65535 && KS <= 65535 && (long)(KS - 1) * kslice < K, "conv3x3_fewout2: split");
    const dim3 grid((unsigned)(tiles_x * tiles_y), KS, B);
    const bool quad = tunable(OODGAN_TUN_FEWOUT_QUAD) != 0;
    const bool al16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(part2)) & 15) == 0;
    if (quad && M == 3 && (!w11t || M2 == 3) && W % 4 == 0 && al16) {
        if (w11t) hipLaunchKernelGGL((conv3x3_fewout3_kernel<true>), grid, dim3(256), 0, as_stream(stream), x, wt, w11t, in_sc, in_sh, part, part2,
                                     K, H, W, kslice, tiles_x);
        else hipLaunchKernelGGL((conv3x3_fewout3_kernel<false>), grid, dim3(256), 0, as_stream(stream), x, wt, w11t, in_sc, in_sh, part, part2,
                                K, H, W, kslice, tiles_x);
    } else if (w11t) hipLaunchKernelGGL((conv3x3_fewout2_kernel<true>), grid, dim3(256), 0, as_stream(stream), x, wt, w11t, in_sc, in_sh, part, part2,
                                 K, M, M2, H, W, kslice, tiles_x);
    else hipLaunchKernelGGL((conv3x3_fewout2_kernel<false>), grid, dim3(256), 0, as_stream(stream), x, wt, w11t, in_sc, in_sh, part, part2,
                            K, M, M2, H, W, kslice, tiles_x);
    const long HW = (long)H * W;
    hipLaunchKernelGGL(conv3x3_fewout_finish_kernel, dim3(stream_grid((long)B * M * HW, 256)), dim3(256), 0, as_stream(stream), part, slope, y,
                       KS, M, HW, (long)B * M * HW);
    if (w11t)
        hipLaunchKernelGGL(conv3x3_fewout_finish_kernel, dim3(stream_grid((long)B * M2 * HW, 256)), dim3(256), 0, as_stream(stream), part2,
                           (const float*)nullptr, y2, KS, M2, HW, (long)B * M2 * HW);
    return check_launch("conv3x3_fewout2");
}

extern "C" int oodgan_align_head(const float* x, float* y, int B, long HW, float scale, void* stream) {
    OODGAN_REQUIRE(x && y && B > 0 && HW > 0, "align_head: bad args");
    const long total = (long)B * 3 * HW;
    hipLaunchKernelGGL(align_head_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), x, y, HW, total, scale);
    return check_launch("align_head");
}

extern "C" int oodgan_field_compose(const float* acc, const float* cur, const float* prev, float* out, int B, int H, int W, int Hp,
                                    int Wp, float scale, int mode, void* stream) {
    OODGAN_REQUIRE(cur && out && B > 0 && H > 0 && W > 0, "field_compose: bad args");
    OODGAN_REQUIRE((mode == 0 && acc) || (mode == 1 && prev && Hp > 0 && Wp > 0), "field_compose: mode %d operands", mode);
    hipLaunchKernelGGL(field_compose_kernel, dim3(stream_grid((long)B * H * W, 256)), dim3(256), 0, as_stream(stream), acc, cur,
                       prev, out, B, H, W, Hp, Wp, scale, mode);
    return check_launch("field_compose");
}

extern "C" int oodgan_warp_blend(const float* target, const float* field, float* y, int B, int C, int H, int W, void* stream) {
    OODGAN_REQUIRE(target && field && y && B > 0 && C > 0 && H > 0 && W > 0, "warp_blend: bad args");
    OODGAN_REQUIRE(B <= 65535 && (C + WB_CPB - 1) / WB_CPB <= 65535, "warp_blend: B or C too large");
    dim3 grid((unsigned)(((long)H * W + 255) / 256), (unsigned)((C + WB_CPB - 1) / WB_CPB), B);
    hipLaunchKernelGGL(warp_blend_kernel, grid, dim3(256), 0, as_stream(stream), target, field, y, C, H, W);
    return check_launch("warp_blend");
}

extern "C" int oodgan_mask_blend(const float* const* fields, const int* sizes, int nfields, const float* x, const float* gen,
                                 float* alpha_out, float* out, int B, int S, void* stream) {
    OODGAN_REQUIRE(fields && sizes && nfields >= 1 && nfields <= 4 && B > 0 && S > 0, "mask_blend: bad args");
    OODGAN_REQUIRE(!out || (x && gen), "mask_blend: out needs x and gen");
    MaskArgs m{};
    m.nfields = nfields;
    for (int i = 0; i < nfields; ++i) {
        OODGAN_REQUIRE(fields[i] && sizes[i] > 0, "mask_blend: field %d", i);
        m.fields[i] = fields[i];
        m.sizes[i] = sizes[i];
    }
    long chunks = ((long)S * S + 255) / 256;
    if (chunks > 1024) chunks = 1024;
    hipLaunchKernelGGL(mask_blend_kernel, dim3((unsigned)chunks, B), dim3(256), 0, as_stream(stream), m, x, gen, alpha_out, out, S);
    return check_launch("mask_blend");
}

extern "C" int oodgan_resize_nearest(const float* x, float* y, int planes, int Hin, int Win, int Hout, int Wout, int out_pitch,
                                     int out_xoff, void* stream) {
    OODGAN_REQUIRE(x && y && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "resize_nearest: bad args");
    if (out_pitch == 0) out_pitch = Wout;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(stream_grid((long)planes * Hout * Wout, 256)), dim3(256), 0, as_stream(stream),
                       x, y, planes, Hin, Win, Hout, Wout, out_pitch, out_xoff);
    return check_launch("resize_nearest");
}

extern "C" int oodgan_resize_bilinear(const float* x, float* y, int planes, int Hin, int Win, int Hout, int Wout, void* stream) {
    OODGAN_REQUIRE(x && y && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "resize_bilinear: bad args");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(stream_grid((long)planes * Hout * Wout, 256)), dim3(256), 0, as_stream(stream),
                       x, y, planes, Hin, Win, Hout, Wout);
    return check_launch("resize_bilinear");
}
