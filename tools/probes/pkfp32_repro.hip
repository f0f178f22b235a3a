// Stand-alone reproducer of DESIGN.md §10 (no torch, no Python): a ToRGB-shaped kernel whose channel loop the compiler turns into
// packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 with op_sel broadcast operands) runs on HIP stream A while a
// matrix-core kernel runs on stream B; every output is compared with the output of the same kernel run alone.
//
//   tools/probes/pkfp32_repro.sh            # builds two binaries from THIS file and runs both, all neighbours
//     build/pkfp32_repro_pk     hipcc --offload-arch=gfx950 -O3                                   (packed fp32 allowed)
//     build/pkfp32_repro_nopk   ... -Xclang -target-feature -Xclang -packed-fp32-ops              (the library's flags)
//
// Neighbours on stream B:
//   strip      conv_f16s_strip_kernel of liboodgan_hip.so through its C ABI (32 -> 32 channels at 512x512, S-form input):
//              one of the culprits identified in round 1 (with conv_f16s_t2v2 / conv_f16s_s2v2)
//   synthetic  a self-contained MFMA + LDS loop defined below (v_mfma_f32_32x32x16_f16 on fragments re-read from LDS)
//   none       nothing on stream B (control: must be 0 corrupted for both builds)
// Output per (build, neighbour): "corrupted N of 600" — round 1 measured 524 of 600 for the packed build beside the library's
// matrix kernels and 0 of 600 without packed-fp32 instructions.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/oodgan.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ToRGB forward of round 1 (1x1 modulated conv Ci -> 3 + bias + FIR-upsampled skip), four pixels per wave, lanes over channels
__global__ __launch_bounds__(256) void torgb_victim(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ s,
                                                    int s_stride, const float* __restrict__ bias, const float* __restrict__ skip,
                                                    const float* __restrict__ kern, float* __restrict__ y, int Ci, int H, int W, float scale) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long HW = (long)H * W;
    const long p = ((long)blockIdx.x * 4 + wave) * 4;
    if (p >= HW) return;
    const float* xp = x + (long)b * Ci * HW + p;
    float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    for (int ci = lane; ci < Ci; ci += 64) {
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
    const int j = lane;
    float o0 = a0[0], o1 = a1[0], o2 = a2[0];
    if (j == 1) { o0 = a0[1]; o1 = a1[1]; o2 = a2[1]; }
    if (j == 2) { o0 = a0[2]; o1 = a1[2]; o2 = a2[2]; }
    if (j == 3) { o0 = a0[3]; o1 = a1[3]; o2 = a2[3]; }
    o0 += bias[0]; o1 += bias[1]; o2 += bias[2];
    const int h2 = H >> 1, w2_ = W >> 1;
    const int Y = (int)((p + j) / W), X = (int)((p + j) % W);
    const float* sp = skip + (long)b * 3 * h2 * w2_;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ky = (Y & 1) + 2 * t;
        const int iy = (Y + ky - 2) >> 1;
        if (Y + ky - 2 < 0 || iy >= h2) continue;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kx = (X & 1) + 2 * u;
            const int ix = (X + kx - 2) >> 1;
            if (X + kx - 2 < 0 || ix >= w2_) continue;
            const float kv = kern[(3 - ky) * 4 + (3 - kx)];
            const long q = (long)iy * w2_ + ix;
            o0 += kv * sp[q];
            o1 += kv * sp[(long)h2 * w2_ + q];
            o2 += kv * sp[2L * h2 * w2_ + q];
        }
    }
    float* yp = y + (long)b * 3 * HW + p + j;
    yp[0] = o0; yp[HW] = o1; yp[2 * HW] = o2;
}

// synthetic neighbour: 4 waves, 64 KB of LDS, fragments re-read from LDS for every matrix instruction
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_neighbour(const uint4* __restrict__ src, float* __restrict__ dst, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[(blockIdx.x * 4096 + i) & 0xffff];
    __syncthreads();
    float16v acc0 = {}, acc1 = {};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint4 a = lds[(wave * 1024 + k * 64 + lane + it * 8) & 4095];
            const uint4 b = lds[(wave * 1024 + 512 + k * 64 + lane + it * 8) & 4095];
            half8 ha, hb;
            __builtin_memcpy(&ha, &a, 16);
            __builtin_memcpy(&hb, &b, 16);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc1, 0, 0, 0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    dst[(long)blockIdx.x * 256 + threadIdx.x] = r;
}

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static float urand() {
    g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((g_rng >> 40) & 0xffffff) / 8388608.0f - 1.0f;       // [-1, 1)
}
static float* dev_random(size_t n, float mul = 1.f, float add = 0.f) {
    std::vector<float> h(n);
    for (auto& v : h) v = add + mul * urand();
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "ood-gan-inversion_amd/oodgan/liboodgan_hip.so";
#ifdef PK_TAG
    const char* tag = PK_TAG;
#else
    const char* tag = "?";
#endif
    const int B = 4, C0 = 64, H0 = 128;
    const long HW = (long)H0 * H0;
    float* x0 = dev_random((size_t)B * C0 * HW);
    float* s0 = dev_random((size_t)B * C0, 0.3f, 1.f);
    float* wr = dev_random(3 * C0);
    float* bias = dev_random(3);
    float* skip = dev_random((size_t)B * 3 * HW / 4);
    const float k1[4] = {1, 3, 3, 1};
    float kh[16];
    for (int i = 0; i < 16; ++i) kh[i] = k1[i / 4] * k1[i % 4] / 64.f * 4.f;
    float* kern;
    CK(hipMalloc(&kern, 64));
    CK(hipMemcpy(kern, kh, 64, hipMemcpyHostToDevice));
    const int NOUT = 60;
    const size_t ybytes = (size_t)B * 3 * HW * 4;
    float* ys;
    CK(hipMalloc(&ys, ybytes * NOUT));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const dim3 grid((unsigned)((HW / 4 + 3) / 4), B);
    auto victim = [&](float* y) {
        hipLaunchKernelGGL(torgb_victim, grid, dim3(256), 0, sa, x0, wr, s0, C0, bias, skip, kern, y, C0, H0, H0, 1.0f / sqrtf((float)C0));
    };
    std::vector<float> ref(ybytes / 4), got(ybytes / 4);
    victim(ys);
    CK(hipStreamSynchronize(sa));
    CK(hipMemcpy(ref.data(), ys, ybytes, hipMemcpyDeviceToHost));

    // ---- neighbour 1: the library's strip conv through the C ABI
    void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    typedef long (*bytes3_t)(int, int, int);
    typedef long (*bytes4_t)(int, int, int, int);
    typedef int (*pack_t)(const float*, void*, float*, int, int, float, int, int, void*);
    typedef int (*tosf_t)(const float*, const float*, int, const float*, int, const float*, void*, int, int, int, int, int, unsigned*, void*);
    typedef int (*conv_t)(const oodgan_conv_args*, const float*, void*);
    typedef const char* (*err_t)(void);
    oodgan_conv_args ca;
    memset(&ca, 0, sizeof(ca));
    float* unscale2 = nullptr;
    conv_t conv = nullptr;
    err_t lasterr = nullptr;
    if (h) {
        const int Bn = 4, K = 32, M = 32, Hn = 512;
        bytes3_t pbytes = (bytes3_t)dlsym(h, "oodgan_pack_conv3x3_f16s_bytes");
        bytes4_t sbytes = (bytes4_t)dlsym(h, "oodgan_sform_bytes");
        pack_t pack = (pack_t)dlsym(h, "oodgan_pack_conv3x3_f16s");
        tosf_t tosf = (tosf_t)dlsym(h, "oodgan_to_sform");
        conv = (conv_t)dlsym(h, "oodgan_conv3x3_f16s");
        lasterr = (err_t)dlsym(h, "oodgan_last_error");
        float* xn = dev_random((size_t)Bn * K * Hn * Hn);
        float* sn = dev_random((size_t)Bn * K, 0.3f, 1.f);
        float* dn = dev_random((size_t)Bn * M, 0.3f, 1.f);
        float* wn = dev_random((size_t)M * K * 9, 1.f / sqrtf(K * 9.f));
        void *wpk, *xs;
        float* yn;
        CK(hipMalloc(&wpk, pbytes(M, K, 0)));
        CK(hipMalloc(&xs, sbytes(Bn, K, Hn, Hn)));
        CK(hipMemset(xs, 0, sbytes(Bn, K, Hn, Hn)));
        CK(hipMalloc(&yn, (size_t)Bn * M * Hn * Hn * 4));
        CK(hipMalloc(&unscale2, 8));
        int rc = pack(wn, wpk, unscale2, M, K, 1.f, 0, 0, sb);
        rc |= tosf(xn, sn, K, nullptr, 0, nullptr, xs, Bn, K, Hn, Hn, Hn, nullptr, sb);
        if (rc) { fprintf(stderr, "library setup failed: %s\n", lasterr()); return 2; }
        ca.x = (const float*)xs; ca.wpk = (const float*)wpk; ca.out_scale = dn; ca.out_scale_stride = M; ca.y = yn;
        ca.B = Bn; ca.K = K; ca.M = M; ca.Hin = Hn; ca.Win = Hn; ca.mode = OODGAN_CONV_S1; ca.act = OODGAN_ACT_NONE; ca.x_sform = 1;
        CK(hipStreamSynchronize(sb));
    } else {
        fprintf(stderr, "%s not loaded (%s): neighbour 'strip' skipped\n", libpath, dlerror());
    }
    // ---- neighbour 2: synthetic
    uint4* nsrc;
    float* ndst;
    CK(hipMalloc(&nsrc, 65536 * 16));
    CK(hipMemset(nsrc, 0x3c, 65536 * 16));
    CK(hipMalloc(&ndst, 1024 * 256 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_neighbour), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));

    for (int nb = 0; nb < 3; ++nb) {
        const char* nname = nb == 0 ? "none" : nb == 1 ? "strip (liboodgan_hip.so)" : "synthetic mfma+lds";
        if (nb == 1 && !conv) continue;
        int bad = 0, tot = 0, badvals = 0, hi_lanes = 0;
        for (int it = 0; it < 10; ++it) {
            for (int r = 0; r < 12; ++r) {
                if (nb == 1) {
                    if (conv(&ca, unscale2, sb)) { fprintf(stderr, "conv failed: %s\n", lasterr()); return 2; }
                } else if (nb == 2) {
                    hipLaunchKernelGGL(mfma_neighbour, dim3(1024), dim3(256), 65536, sb, nsrc, ndst, 400);
                }
            }
            for (int o = 0; o < NOUT; ++o) victim(ys + (size_t)o * (ybytes / 4));
            CK(hipDeviceSynchronize());
            for (int o = 0; o < NOUT; ++o) {
                CK(hipMemcpy(got.data(), ys + (size_t)o * (ybytes / 4), ybytes, hipMemcpyDeviceToHost));
                int nbad = 0;
                for (size_t i = 0; i < got.size(); ++i)
                    if (memcmp(&got[i], &ref[i], 4)) ++nbad;
                bad += nbad != 0;
                badvals += nbad;
                ++tot;
            }
        }
        printf("build %-5s neighbour %-26s corrupted %d of %d launches (%d wrong values)\n", tag, nname, bad, tot, badvals);
        fflush(stdout);
    }
    return 0;
}
