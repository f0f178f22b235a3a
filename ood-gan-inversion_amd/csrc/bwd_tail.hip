// Batched tail of the W+ backward: the per-layer reductions of the style-gradient partials, the demodulation gradient and
// the range-scale checks are all independent of the gradient chain through the feature maps — they only feed the
// (B, sum Ci) style-gradient accumulator that style_affine_bwd consumes at the very end.  Launched per layer they are
// ~75 kernels of 5-9 us each per step; here each kind is ONE launch whose job table travels in the kernel arguments.
// Arithmetic and summation order per output are those of reduce_parts_kernel / demod_bwd_kernel /
// absmax_scale_check_kernel (autograd of ModulatedConv2d's modulation + demodulation, model.py:236-241).
#include "common.hpp"

using namespace oodgan;

namespace {

constexpr int kMaxReduce = 48, kMaxDemod = 36, kMaxCheck = 36;

struct ReduceTable {
    oodgan_reduce_job j[kMaxReduce];
    int first_wave[kMaxReduce + 1];
    int n;
};

// rows with at most 64 partials (every layer below 128²) take 16 lanes each, four rows per wave: one wave per row made the launch
// 69 000 waves of one or two loads — bound by the rate at which waves start (41 us)
__host__ __device__ __forceinline__ bool reduce_quad(const oodgan_reduce_job& q) { return q.nparts <= 64 && (!q.part2 || q.nparts2 <= 64); }

__global__ __launch_bounds__(256) void reduce_batch_kernel(const ReduceTable t) {
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= t.first_wave[t.n]) return;
    int k = 0;
    while (k + 1 < t.n && wv >= t.first_wave[k + 1]) ++k;
    const oodgan_reduce_job& q = t.j[k];
    const int lane = threadIdx.x & 63;
    const bool quad = reduce_quad(q);
    const long nrows = (long)q.B * q.C;
    const long row_raw = quad ? (wv - t.first_wave[k]) * 4 + (lane >> 4) : wv - t.first_wave[k];
    const bool live = row_raw < nrows;
    const long row = live ? row_raw : nrows - 1;
    // quad: 16 lanes per row, four rows per wave; otherwise one row per wave — the order of a row's sum depends only on its length
    float s = quad ? row_sum16(q.part + row * q.nparts, q.nparts, lane & 15) : row_sum(q.part + row * q.nparts, q.nparts, lane);
    if (q.part2) {
        const float* p2 = q.part2 + row * q.nparts2;
        const float s2 = quad ? row_sum16(p2, q.nparts2, lane & 15) : row_sum(p2, q.nparts2, lane);
        s += q.scale2[(row / q.C) * q.scale2_stride + (row % q.C)] * s2;
    }
    if (quad) {
        if (live && (lane & 15) == 0) {
            float* o = q.out + (row / q.C) * q.out_stride + (row % q.C);
            *o = q.accumulate ? *o + s : s;
        }
        return;
    }
    if (lane == 0) {
        float* o = q.out + (row / q.C) * q.out_stride + (row % q.C);
        *o = q.accumulate ? *o + s : s;
    }
}

struct DemodTable {
    oodgan_demod_bwd_job j[kMaxDemod];
    int first_block[kMaxDemod + 1];
    int n;
};

__global__ __launch_bounds__(256) void demod_bwd_batch_kernel(const DemodTable t) {
    __shared__ float rd2[1024];
    __shared__ float red[4][64];
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
    const oodgan_demod_bwd_job& q = t.j[k];
    const int lb = blockIdx.x - t.first_block[k];
    const int nbx = (q.Ci + 63) / 64;
    const int b = lb / nbx;
    const int cil = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int ci = (lb % nbx) * 64 + cil;
    float acc = 0.f;
    for (int c0 = 0; c0 < q.Co; c0 += 1024) {
        __syncthreads();
        for (int e = threadIdx.x; e < 1024 && c0 + e < q.Co; e += 256) {
            const float dv = q.d[(long)b * q.d_stride + c0 + e];
            rd2[e] = q.r[(long)b * q.Co + c0 + e] * dv * dv;
        }
        __syncthreads();
        const int cn = (q.Co - c0) < 1024 ? (q.Co - c0) : 1024;
        if (ci < q.Ci) {
#pragma unroll 8
            for (int co = cg; co < cn; co += 4) acc += rd2[co] * q.wsq[(long)(c0 + co) * q.Ci + ci];
        }
    }
    red[cg][cil] = acc;
    __syncthreads();
    if (cg == 0 && ci < q.Ci)
        q.gs[(long)b * q.gs_stride + ci] += -(q.scale * q.scale) * q.s[(long)b * q.s_stride + ci] *
                                            (red[0][cil] + red[1][cil] + red[2][cil] + red[3][cil]);
}

struct DemodFwdTable {
    oodgan_demod_fwd_job j[kMaxDemod];
    int first_wave[kMaxDemod + 1];
    int n;
};

// demod_fwd_kernel for every styled conv of the generator in one launch (all of them only need the style vector)
__global__ __launch_bounds__(256) void demod_fwd_batch_kernel(const DemodFwdTable t) {
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= t.first_wave[t.n]) return;
    int k = 0;
    while (k + 1 < t.n && wv >= t.first_wave[k + 1]) ++k;
    const oodgan_demod_fwd_job& q = t.j[k];
    const long idx = wv - t.first_wave[k];
    const int lane = threadIdx.x & 63;
    const int b = (int)(idx / q.Co), co = (int)(idx % q.Co);
    const float acc = demod_dot(q.s + (long)b * q.s_stride, q.wsq + (long)co * q.Ci, q.Ci, lane);
    if (lane == 0) q.d[(long)b * q.d_stride + co] = rsqrtf(acc * (q.scale * q.scale) + 1e-8f);
}

struct CheckTable {
    oodgan_scale_check_job j[kMaxCheck];
    int n;
};

__global__ __launch_bounds__(1024) void scale_check_batch_kernel(const CheckTable t, int* __restrict__ flag) {
    __shared__ float red[16];
    const oodgan_scale_check_job& q = t.j[blockIdx.x];
    float m = 0.f;
    bool bad = false;
    for (long i = threadIdx.x; i < q.n; i += 4096) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i + 1024 * u < q.n ? q.part[i + 1024 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!isfinite(v[u])) bad = true;
            m = fmaxf(m, fabsf(v[u]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 2);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
        const float used = q.state[1];
        if (m > 0.f && isfinite(m)) {
            const float scaled = m * used;
            if (!(scaled >= 0.00390625f && scaled < 32768.f)) atomicOr(flag, 1);
        }
        int e = 0;
        if (m > 0.f && isfinite(m)) e = 9 - (int)floorf(log2f(m));
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        q.state[0] = ldexpf(1.f, -e);
        q.state[1] = ldexpf(1.f, e);
    }
}

}  // namespace

extern "C" int oodgan_reduce_batch(const oodgan_reduce_job* jobs, int njobs, void* stream) {
    OODGAN_REQUIRE(jobs && njobs > 0, "reduce_batch: bad args");
    for (int base = 0; base < njobs; base += kMaxReduce) {
        ReduceTable t;
        t.n = njobs - base < kMaxReduce ? njobs - base : kMaxReduce;
        long waves = 0;
        for (int i = 0; i < t.n; ++i) {
            const oodgan_reduce_job& q = jobs[base + i];
            OODGAN_REQUIRE(q.part && q.out && q.B > 0 && q.C > 0 && q.nparts > 0 && q.out_stride >= q.C, "reduce_batch: bad job %d", base + i);
            t.j[i] = q;
            t.first_wave[i] = (int)waves;
            waves += reduce_quad(q) ? ((long)q.B * q.C + 3) / 4 : (long)q.B * q.C;
        }
        OODGAN_REQUIRE(waves < (1L << 31), "reduce_batch: too many rows");
        t.first_wave[t.n] = (int)waves;
        hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), t);
        int rc = check_launch("reduce_batch");
        if (rc != OODGAN_OK) return rc;
    }
    return OODGAN_OK;
}

extern "C" int oodgan_demod_bwd_batch(const oodgan_demod_bwd_job* jobs, int njobs, void* stream) {
    OODGAN_REQUIRE(jobs && njobs > 0, "demod_bwd_batch: bad args");
    for (int base = 0; base < njobs; base += kMaxDemod) {
        DemodTable t;
        t.n = njobs - base < kMaxDemod ? njobs - base : kMaxDemod;
        long blocks = 0;
        for (int i = 0; i < t.n; ++i) {
            const oodgan_demod_bwd_job& q = jobs[base + i];
            OODGAN_REQUIRE(q.s && q.wsq && q.d && q.r && q.gs && q.B > 0 && q.Ci > 0 && q.Co > 0, "demod_bwd_batch: bad job %d", base + i);
            t.j[i] = q;
            t.first_block[i] = (int)blocks;
            blocks += (long)((q.Ci + 63) / 64) * q.B;
        }
        t.first_block[t.n] = (int)blocks;
        hipLaunchKernelGGL(demod_bwd_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), t);
        int rc = check_launch("demod_bwd_batch");
        if (rc != OODGAN_OK) return rc;
    }
    return OODGAN_OK;
}

extern "C" int oodgan_absmax_scale_check_batch(const oodgan_scale_check_job* jobs, int njobs, int* flag, void* stream) {
    OODGAN_REQUIRE(jobs && njobs > 0 && flag, "absmax_scale_check_batch: bad args");
    for (int base = 0; base < njobs; base += kMaxCheck) {
        CheckTable t;
        t.n = njobs - base < kMaxCheck ? njobs - base : kMaxCheck;
        for (int i = 0; i < t.n; ++i) {
            OODGAN_REQUIRE(jobs[base + i].part && jobs[base + i].state && jobs[base + i].n > 0, "absmax_scale_check_batch: bad job %d", base + i);
            t.j[i] = jobs[base + i];
        }
        hipLaunchKernelGGL(scale_check_batch_kernel, dim3(t.n), dim3(1024), 0, as_stream(stream), t, flag);
        int rc = check_launch("absmax_scale_check_batch");
        if (rc != OODGAN_OK) return rc;
    }
    return OODGAN_OK;
}

extern "C" int oodgan_demod_fwd_batch(const oodgan_demod_fwd_job* jobs, int njobs, void* stream) {
    OODGAN_REQUIRE(jobs && njobs > 0, "demod_fwd_batch: bad args");
    for (int base = 0; base < njobs; base += kMaxDemod) {
        DemodFwdTable t;
        t.n = njobs - base < kMaxDemod ? njobs - base : kMaxDemod;
        long waves = 0;
        for (int i = 0; i < t.n; ++i) {
            const oodgan_demod_fwd_job& q = jobs[base + i];
            OODGAN_REQUIRE(q.s && q.wsq && q.d && q.B > 0 && q.Ci > 0 && q.Co > 0, "demod_fwd_batch: bad job %d", base + i);
            t.j[i] = q;
            t.first_wave[i] = (int)waves;
            waves += (long)q.B * q.Co;
        }
        t.first_wave[t.n] = (int)waves;
        hipLaunchKernelGGL(demod_fwd_batch_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), t);
        int rc = check_launch("demod_fwd_batch");
        if (rc != OODGAN_OK) return rc;
    }
    return OODGAN_OK;
}
