// Forward "producer" of the up-sampling StyledConv tail: Blur(pad=(1,1)) of the transposed-conv output
// (reference src/ops/StyleGAN/model.py:72-88,199-205 -> upfirdn2d(kernel*4, pad=(1,1))), NoiseInjection, bias and
// leaky-ReLU*sqrt2 (model.py:283-292,343-350) in ONE pass that writes
//   * y  (B,C,2H,2W) fp32 — the saved activation (needed by the W+ backward and by ToRGB), and
//   * ys — the same values times the next layer's style, split hi/lo, as that conv's S-form input (sform.hpp),
// so the separate fp32 -> S-form conversion pass (one more read of the tensor) disappears.
// Tiling as blurT_sp_kernel: block = (b, 16-channel block) x an 8 x 64 output tile; the input tile goes to LDS once,
// the 4x4 FIR runs from registers (rank-1 kernels as a vertical then a horizontal pass), results are exchanged through
// LDS so that one thread owns one pixel with its 16 channels.
#include "common.hpp"
#include "sform.hpp"
#include <cstdint>

using namespace oodgan;

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int FT_R = 11, FT_C = 72;      // input tile: rows Y0-1..Y0+9, cols X0-4..X0+67
// LDS pitches (floats): row 74, channel plane 816 — with the FIR's lane mapping (4 channels x 8 rows x 2 half rows per wave)
// every ds_read_b64 of a half wave then covers the 64 banks once (72 / 792 was a 4-way conflict); rows stay 8-byte aligned.
constexpr int FT_P = 74, FT_Q = 816;
constexpr int LS_ROW = 65, LS_CH = 520;  // exchange buffer [16 ch][8 rows x 65]: conflict-free scalar writes and row reads

// element k (0..12) of thread i (0..15 within its channel) of the 11-row x 18-float4 input tile: the thread walks down
// column group i (k = row), then rows 0..10 of the two halo column groups 16 and 17 go to threads 0..10 — compile-time
// row offsets and a fixed column per thread instead of a division per element.
__device__ __forceinline__ bool tile_elem(int k, int i, int& r, int& c4) {
    if (k < FT_R) { r = k; c4 = i; return true; }
    r = i; c4 = 16 + (k - FT_R);
    return i < FT_R;
}

struct BlurArgs {
    const float* z;          // (B,C,Hz,pitch), Hz = 2H+1, valid width Wz = 2W+1
    const float* kern;       // 4x4 (the op flips it, as upfirdn2d does)
    float* y;                // (B,C,2H,2W)
    uint4* ys;               // S-form of y*ys_scale or null
    const float* ys_scale;   // (B,*) stride ys_scale_stride or null
    const float* bias;       // (C) or null
    const float* noise;      // (noise_batch,2H,2W) or null
    const float* noise_w;
    int ys_scale_stride, noise_batch, act;
    int B, C, H, W, pitch;
    int tiles_x, tiles_y;
    SDims yd;
    unsigned* vmax;          // (B) or null: max |value written to ys| per sample (fwd_range.hip)
    int y_fform;             // 1: y is written in F-form ([B][C/16][2H][2W][16]) and there is no S-form output; vmax = max |y*ys_scale|
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void blur_act_sform_kernel(const BlurArgs a) {
    __shared__ __attribute__((aligned(16))) float lin[16 * FT_Q];     // reused as lst[16][LS_CH]
    __shared__ float kf[16];
    __shared__ float ksep[9];
    __shared__ float cb[2][16];          // bias, ys_scale of the block's channels
    const int tid = threadIdx.x;
    int w;
    {   // contiguous chunk of the tile list per XCD (halo rows/cols are the neighbours' interiors)
        const int total = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = w % a.tiles_x; w /= a.tiles_x;
    const int ty = w % a.tiles_y; w /= a.tiles_y;
    const int KC = a.yd.KC;
    const int kc = w % KC, b = w / KC;
    const int Y0 = ty * 8, X0 = tx * 64;
    const int Ho = 2 * a.H, Wo = 2 * a.W, Hz = Ho + 1, Wz = Wo + 1;
    if (tid < 16) {
        kf[tid] = a.kern[15 - tid];          // flipped: kf[ky][kx] = k[3-ky][3-kx]
        const int c = kc * 16 + tid;
        cb[0][tid] = (a.bias && c < a.C) ? a.bias[c] : 0.f;
        cb[1][tid] = (a.ys_scale && c < a.C) ? a.ys_scale[(long)b * a.ys_scale_stride + c] : (c < a.C ? 1.f : 0.f);
    }
    __syncthreads();
    if (tid == 0) {
        bool ok = kf[0] != 0.f;
        for (int i = 0; i < 4 && ok; ++i) { ksep[i] = kf[i * 4] / kf[0]; ksep[4 + i] = kf[i]; }
        for (int i = 0; i < 16 && ok; ++i) ok = fabsf(ksep[i >> 2] * ksep[4 + (i & 3)] - kf[i]) <= 1e-7f * fabsf(kf[0]);
        ksep[8] = ok ? 1.f : 0.f;
    }
    // ---- A: 16 threads per channel sweep its 11 x 18 float4 tile (zero outside the (2H+1)x(2W+1) support)
    {
        const int ch = tid >> 4, c = kc * 16 + ch;
        const float* zp = a.z + ((long)b * a.C + c) * Hz * a.pitch;
        // all loads of the sweep are issued before the first LDS store: one dependent load per iteration would make
        // the tile fill a chain of 13 HBM latencies
        constexpr int NE = FT_R + 2;       // 13 (tile_elem)
        float4 v[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            int r, c4;
            const bool act = tile_elem(k, tid & 15, r, c4);
            const int gy = Y0 - 1 + r, gx = X0 - 4 + 4 * c4;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act && c < a.C && gy >= 0 && gy < Hz && gx >= 0 && gx + 3 < a.pitch)
                v[k] = *reinterpret_cast<const float4*>(zp + (long)gy * a.pitch + gx);
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            int r, c4;
            if (!tile_elem(k, tid & 15, r, c4)) continue;
            const int gx = X0 - 4 + 4 * c4;
            if (gx + 3 >= Wz) {          // columns between the valid width and the pitch are not defined
                if (gx + 0 >= Wz) v[k].x = 0.f;
                if (gx + 1 >= Wz) v[k].y = 0.f;
                if (gx + 2 >= Wz) v[k].z = 0.f;
                v[k].w = 0.f;
            }
            float2* dst = reinterpret_cast<float2*>(lin + ch * FT_Q + r * FT_P + 4 * c4);
            dst[0] = make_float2(v[k].x, v[k].y);
            dst[1] = make_float2(v[k].z, v[k].w);
        }
    }
    __syncthreads();
    // ---- B: out[Y][X] = sum kf[ky][kx] z[Y-1+ky][X-1+kx]; thread = (channel, output row, half row of 32)
    const int ch = tid >> 4, tq = tid & 15;
    const int yrow = tq >> 1, xh = tq & 1;
    float o[32];
    if (ksep[8] != 0.f) {
        // two halves of 16 outputs: a 20-wide vertical pass each (keeps the live set at 32 + 20 registers)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float tmp[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) tmp[j] = 0.f;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                // tile column of X-1 is (X - X0) + 3; read the aligned pairs from column 32*xh + 16*hf + 2 and skip one
                const float2* row = reinterpret_cast<const float2*>(lin + ch * FT_Q + (yrow + aa) * FT_P + 32 * xh + 16 * hf + 2);
                const float kv = ksep[aa];
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const float2 v = row[j];
                    tmp[2 * j] += kv * v.x;
                    tmp[2 * j + 1] += kv * v.y;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                o[16 * hf + j] = ksep[4] * tmp[j + 1] + ksep[5] * tmp[j + 2] + ksep[6] * tmp[j + 3] + ksep[7] * tmp[j + 4];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) o[j] = 0.f;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            const float* row = lin + ch * FT_Q + (yrow + aa) * FT_P + 32 * xh + 3;
            float win[35];
#pragma unroll
            for (int j = 0; j < 35; ++j) win[j] = row[j];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const float kv = kf[aa * 4 + bb];
#pragma unroll
                for (int j = 0; j < 32; ++j) o[j] += kv * win[j + bb];
            }
        }
    }
    __syncthreads();
    float* lst = lin;
#pragma unroll
    for (int j = 0; j < 32; ++j) lst[ch * LS_CH + yrow * LS_ROW + 32 * xh + j] = o[j];
    __syncthreads();
    // ---- C: thread = pixel: noise + bias + activation for its 16 channels, fp32 store per channel plane; the value for the
    // S-form (x the next conv's style) goes back to the same LDS cell
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const long HWo = (long)Ho * Wo;
    float vm = 0.f;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int pos = tid + rep * 256;
        const int Y = Y0 + (pos >> 6), X = X0 + (pos & 63);
        if (Y >= Ho || X >= Wo) continue;
        const float nz = a.noise ? nw * a.noise[(long)(a.noise_batch > 1 ? b : 0) * HWo + (long)Y * Wo + X] : 0.f;
        float* yp = a.y + ((long)b * a.C + kc * 16) * HWo + (long)Y * Wo + X;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            float* cell = &lst[cc * LS_CH + (pos >> 6) * LS_ROW + (pos & 63)];
            float t = *cell + nz + cb[0][cc];
            if (a.act == OODGAN_ACT_LRELU) t = (t > 0.f ? t : 0.2f * t) * kSqrt2;
            if (!a.y_fform && kc * 16 + cc < a.C) yp[(long)cc * HWo] = t;
            const float v = t * cb[1][cc];
            vm = fmaxf(vm, fabsf(v));
            *cell = a.y_fform ? t : v;
        }
    }
    if (a.y_fform) {
        // ---- D': the 64-byte fp32 records, one channel quarter (16 bytes) per thread and pass, the 64 lanes of a store writing 64
        // consecutive quarters (1 KB contiguous)
        __syncthreads();
        float* yf = a.y + ((long)b * KC + kc) * HWo * 16;
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            const int u = tid + rep * 256;
            const int pos = u >> 2, q = u & 3;
            const int Y = Y0 + (pos >> 6), X = X0 + (pos & 63);
            if (Y >= Ho || X >= Wo) continue;
            const float* cell = &lst[(4 * q) * LS_CH + (pos >> 6) * LS_ROW + (pos & 63)];
            *reinterpret_cast<float4*>(yf + ((long)Y * Wo + X) * 16 + 4 * q) = make_float4(cell[0], cell[LS_CH], cell[2 * LS_CH], cell[3 * LS_CH]);
        }
    }
    if (a.ys && !a.y_fform) {
        // ---- D: the 64-byte records, one 16-byte slot per thread and pass (slots 0,1 = hi halves of channels 0-7 / 8-15, slots
        // 2,3 = lo halves), ordered so that the 64 lanes of a store write 64 CONSECUTIVE slots (1 KB contiguous): one thread per
        // record issues four 16-byte pieces at a 64-byte stride, four times the memory transactions
        __syncthreads();
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            const int u = tid + rep * 256;
            const int pos = u >> 2, sl = u & 3;
            const int Y = Y0 + (pos >> 6), X = X0 + (pos & 63);
            if (Y >= Ho || X >= Wo) continue;
            half8 o8;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const float val = lst[(8 * (sl & 1) + cc) * LS_CH + (pos >> 6) * LS_ROW + (pos & 63)];
                const _Float16 hh = (_Float16)val;
                o8[cc] = (sl & 2) ? (_Float16)(val - (float)hh) : hh;
            }
            reinterpret_cast<half8*>(a.ys + sform_unit(a.yd, b, kc, Y, X, 0))[sl] = o8;
        }
    }
    if (a.vmax) record_vmax(a.vmax, b, vm);
}


// ---------------------------------------------------------------------------------------------------------------
// STRIP WALK of the same tail for the F-form output (the 1024² level inside the W+ loop, oodgan_blur_act_fform): the tile kernel
// above fetches an 11 x 72 input tile for every 8 x 64 outputs and loads only during the first of its four phases (623 us for
// 2.15 GB at B = 8).  Here a workgroup owns (b, 16-channel block, 64 output columns) and walks DOWN a segment of output rows:
//   * thread = (channel, column quad); one iteration = one output row Y = one NEW z row Y + 2, requested three iterations earlier
//     (register prefetch: a float4 per thread, the three halo columns by the edge threads of a channel, a float4 of noise);
//   * the kernel must be rank one (Blur's [1,3,3,1] x [1,3,3,1] is; the caller falls back to the tile kernel otherwise): one
//     horizontal 4-tap pass of the new row — the neighbours' columns come by DPP inside the channel's 16 lanes — then the
//     vertical pass over the last four filtered rows kept in registers;
//   * noise + bias + activation, the 16 channels of a pixel meet through a double-buffered LDS image (one barrier per row) and
//     leave as 64-byte fp32 records, 16 bytes per lane, 1 KiB contiguous per wave.
struct BlurStripGeo { int nstrips, nseg, seg_rows; };

// SF: instead of the F-form records the kernel writes y as fp32 planes (a float4 per thread and row: its channel's four columns)
// and the S-form of y * ys_scale (the gather image then holds the scaled values; a thread converts the 8 channels of its slot).
// FULL: 2W % 64 == 0 — every lane of every strip stores, so the stores (and the noise load, from a valid address when there is no
// noise) sit in straight-line code: the compiler can count what is in flight only through code without branches around memory
// operations; with an `if` around a store it drained everything (vmcnt(0)) twice per three rows.
template <bool SF, bool FULL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void blur_act_fform_strip_kernel(const BlurArgs a, const BlurStripGeo geo) {
    constexpr int GP = 68;                                   // LDS pitch of a channel's 64 columns
    __shared__ __attribute__((aligned(16))) float gat[2][16][GP];
    const int tid = threadIdx.x;
    int w;
    {
        const int total = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int strip = w % geo.nstrips; w /= geo.nstrips;
    const int seg = w % geo.nseg; w /= geo.nseg;
    const int KC = a.yd.KC, kc = w % KC, b = w / KC;
    const int Ho = 2 * a.H, Wo = 2 * a.W, Hz = Ho + 1, Wz = Wo + 1;
    const int Y0 = seg * geo.seg_rows, Y1 = min(Y0 + geo.seg_rows, Ho);
    const int X0 = 64 * strip;
    // kf[ky][kx] = k[3-ky][3-kx] (upfirdn2d flips); rank one: kf[ky][kx] = kv[ky] * kh[kx] with kv[ky] = kf[ky][j] / kf[i][j], kh = kf[i][.] for a pivot (i, j)
    // (uniform loads, no LDS table and no barrier in front of the first row request)
    // The pivot is the corner tap when it is non-zero (the usual case, and the one the parity tests pin bit for bit); a rank-one kernel
    // with a zero corner ([0,1,1,0] x [0,1,1,0]) takes its largest tap as pivot instead — no division by zero, no NaN.
    int pr = 3, pc = 3;
    float k00 = a.kern[15];
    if (!(fabsf(k00) > 0.f)) {
        float best = 0.f;
        for (int i = 0; i < 16; ++i) {
            const float v = fabsf(a.kern[i]);
            if (v > best) { best = v; pr = i >> 2; pc = i & 3; }
        }
        k00 = a.kern[pr * 4 + pc];
    }
    const bool piv = fabsf(k00) > 0.f;
    const float kv0 = piv ? a.kern[12 + pc] / k00 : 0.f, kv1 = piv ? a.kern[8 + pc] / k00 : 0.f, kv2 = piv ? a.kern[4 + pc] / k00 : 0.f,
                kv3 = piv ? a.kern[pc] / k00 : 0.f;
    const float kh0 = a.kern[pr * 4 + 3], kh1 = a.kern[pr * 4 + 2], kh2 = a.kern[pr * 4 + 1], kh3 = a.kern[pr * 4];
    const int ch = tid >> 4, q = tid & 15, c = kc * 16 + ch;
    const float* zp = a.z + ((long)b * a.C + c) * Hz * a.pitch;
    const float bv = a.bias ? a.bias[c] : 0.f;
    const float ysc = a.ys_scale ? a.ys_scale[(long)b * a.ys_scale_stride + c] : 1.f;
    const float nw = a.noise ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    const long HWo = (long)Ho * Wo;
    const float* np = a.noise ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * HWo : nullptr;
    const int gx = X0 + 4 * q;                               // this thread's output columns gx .. gx+3 = z columns gx-1 .. gx+5 are needed
    // one z row: own float4 (columns gx .. gx+3) and the halo of the edge threads (q = 0: column X0-1; q = 15: columns X0+64, X0+65).
    // A load is ONLY a load here: every lane reads from a clamped address, and what lies outside the (2H+1) x (2W+1) support
    // (columns between the valid width and the pitch are not defined) is masked where the row is CONSUMED, three iterations later,
    // with lane masks that do not depend on the row.  Masking a value right behind its load (`if (gx >= Wz) m.x = 0`) is a use: the
    // compiler waits for the load — vmcnt(0) — at that point, in every iteration, and the rows in flight are worth nothing (that
    // form of this kernel: 76 % of the wave time waiting at 3.5 TB/s).  For the same reason the three row buffers rotate by NAME
    // (the loop body is written three times): `R0 = R1` is a read of R1.
    struct Row { float4 m; float e0, e1; float4 n; float rv; };
    const int gxc = min(gx, a.pitch - 4);                    // gx + 3 >= pitch <=> gx >= pitch >= Wz: all four columns masked
    const bool mk0 = gx + 0 < Wz, mk1 = gx + 1 < Wz, mk2 = gx + 2 < Wz, mk3 = gx + 3 < Wz;
    const bool q0 = q == 0, q15 = q == 15;
    const int ec0 = q0 ? max(X0 - 1, 0) : min(X0 + 64, a.pitch - 1), ec1 = min(X0 + 65, a.pitch - 1);
    const bool me0 = q0 ? X0 >= 1 : X0 + 64 < Wz, me1 = X0 + 65 < Wz;
    const bool mkn = np != nullptr && gx + 3 < Wo;
    const int gxn = min(gx, Wo - 4);
    const float* nsrc = np ? np : zp;                        // no noise: any valid address (2H x 2W floats lie inside the z plane)
    auto load_row = [&](int r, int Y, Row& R) {
        const float* rp = zp + (long)min(max(r, 0), Hz - 1) * a.pitch;
        R.rv = (r >= 0 && r < Hz) ? 1.f : 0.f;
        R.m = *reinterpret_cast<const float4*>(rp + gxc);    // plain loads and stores: with `nt` hints (common.hpp, ld4_stream) this
        R.e0 = rp[ec0];                                      // kernel is 7-25 % SLOWER at every level (profiles/r4_ab_nt_swizzle.txt)
        R.e1 = rp[ec1];
        R.n = *reinterpret_cast<const float4*>(nsrc + (long)min(max(Y, 0), Ho - 1) * Wo + gxn);
    };
    // horizontal pass of a row: h[e] = sum_b kh[b] * z[gx + e - 1 + b]
    auto hpass = [&](const Row& R, float (&h)[4]) {
        const float mx = mk0 ? R.m.x : 0.f, my = mk1 ? R.m.y : 0.f, mz = mk2 ? R.m.z : 0.f, mw = mk3 ? R.m.w : 0.f;
        const float e0 = me0 ? R.e0 : 0.f, e1 = me1 ? R.e1 : 0.f;
        // left neighbour's last column, right neighbour's first two (DPP row_shr / row_shl inside the channel's 16 lanes)
        float lw = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mw), 0x111, 0xF, 0xF, false));   // row_shr:1
        float rx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x101, 0xF, 0xF, false));   // row_shl:1
        float ry = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, my), 0x101, 0xF, 0xF, false));
        if (q0) lw = e0;
        if (q15) { rx = e0; ry = e1; }
        const float x[7] = {lw, mx, my, mz, mw, rx, ry};
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (kh0 * x[e] + kh1 * x[e + 1] + kh2 * x[e + 2] + kh3 * x[e + 3]) * R.rv;
    };

    // out[Y] = kv0 h[Y-1] + kv1 h[Y] + kv2 h[Y+1] + kv3 h[Y+2]: three filtered rows of history, the new one arrives each iteration
    float h0[4], h1[4], h2[4];
    Row R0, R1, R2;                                          // rows in flight: z rows Y+2, Y+3, Y+4 of the current iteration
    R0.n = R1.n = R2.n = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        Row T0, T1, T2;                                      // all six rows of the prologue are requested before the first is used
        T0.n = T1.n = T2.n = make_float4(0.f, 0.f, 0.f, 0.f);
        load_row(Y0 - 1, -1, T0);
        load_row(Y0, -1, T1);
        load_row(Y0 + 1, -1, T2);
        load_row(Y0 + 2, Y0, R0);
        load_row(Y0 + 3, Y0 + 1, R1);
        load_row(Y0 + 4, Y0 + 2, R2);
        hpass(T0, h0);
        hpass(T1, h1);
        hpass(T2, h2);
    }
    float vm = 0.f;
    float* yf = a.y + ((long)b * KC + kc) * HWo * 16;
    float* yplane = a.y + ((long)b * a.C + min(c, a.C - 1)) * HWo;          // SF: this thread's channel plane
    const bool yok = c < a.C && gx + 3 < Wo;
    const int pcol = tid >> 2, pq = tid & 3;                 // record role: pixel column, channel quarter (F-form) / slot (S-form)
    // Gather image: channels 8-15 keep their columns XOR 8 (the two 8-column halves of every 16 swapped).  ds_read_b32 is served in
    // two groups of 32 lanes with bank = dword address mod 32 (MI355X_MICROARCH.md, LDS): with the 68-dword pitch channel rows c and
    // c + 8 start on the same bank, and the 32 lanes of a group read 8 consecutive columns of channels {q, q + 8} (F-form quarters
    // 0 / 2 and 1 / 3; S-form halves) — a 2-way conflict on every read (30-39 % of the LDS-active cycles in round 3's counters).
    // With the swap the 32 lanes touch 32 different banks; the float4 writes (8 contiguous lanes per group) stay conflict free.
    const int sw = ch & 8;
    const bool pok = X0 + pcol < Wo;
    auto step = [&](int Y, Row& R) __attribute__((always_inline)) {
        // the three steps of a loop trip are one basic block: without a fence the scheduler gathers the masking of all three rows
        // at its top, and the wait for the youngest request (vmcnt(0)) with it
        __builtin_amdgcn_sched_barrier(0);
        float h3[4];
        hpass(R, h3);
        const float nn[4] = {mkn ? R.n.x : 0.f, mkn ? R.n.y : 0.f, mkn ? R.n.z : 0.f, mkn ? R.n.w : 0.f};
        load_row(Y + 5, Y + 3, R);
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = kv0 * h0[e] + kv1 * h1[e] + kv2 * h2[e] + kv3 * h3[e] + nw * nn[e] + bv;
            if (a.act == OODGAN_ACT_LRELU) v = (v > 0.f ? v : 0.2f * v) * kSqrt2;
            t[e] = v;
            if (gx + e < Wo) vm = fmaxf(vm, fabsf(v * ysc));
            h0[e] = h1[e]; h1[e] = h2[e]; h2[e] = h3[e];
        }
        float (*gb)[GP] = gat[Y & 1];
        if constexpr (SF) {
            if (FULL || yok) *reinterpret_cast<float4*>(yplane + (long)Y * Wo + gx) = make_float4(t[0], t[1], t[2], t[3]);
            *reinterpret_cast<float4*>(&gb[ch][(4 * q) ^ sw]) = make_float4(t[0] * ysc, t[1] * ysc, t[2] * ysc, t[3] * ysc);
            __syncthreads();
            if (FULL || pok) {
                // slot pq of the pixel's record: hi (pq < 2) or lo halves of channels 8 (pq & 1) .. + 7; the lanes of a wave write 16 whole records
                half8 o8;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    const float val = gb[8 * (pq & 1) + cc][pcol ^ (8 * (pq & 1))];
                    const _Float16 hh = (_Float16)val;
                    o8[cc] = (pq & 2) ? (_Float16)(val - (float)hh) : hh;
                }
                reinterpret_cast<half8*>(a.ys + sform_unit(a.yd, b, kc, Y, X0 + pcol, 0))[pq] = o8;
            }
        } else {
            *reinterpret_cast<float4*>(&gb[ch][(4 * q) ^ sw]) = make_float4(t[0], t[1], t[2], t[3]);
            __syncthreads();
            if (FULL || pok) {
                const int pc_ = pcol ^ ((pq & 2) << 2);            // channels 8-15 (pq >= 2): column ^ 8
                *reinterpret_cast<float4*>(yf + ((long)Y * Wo + X0 + pcol) * 16 + 4 * pq) =
                    make_float4(gb[4 * pq][pc_], gb[4 * pq + 1][pc_], gb[4 * pq + 2][pc_], gb[4 * pq + 3][pc_]);
            }
        }
    };
    int Y = Y0;
    for (; Y + 2 < Y1; Y += 3) {
        step(Y, R0);
        step(Y + 1, R1);
        step(Y + 2, R2);
    }
    if (Y < Y1) step(Y, R0);
    if (Y + 1 < Y1) step(Y + 1, R1);
    if (a.vmax && c < a.C) record_vmax(a.vmax, b, vm);
}

}  // namespace

static int blur_act_launch(const float* z, const float* kernel, float* y, void* ys, const float* ys_scale,
                           int ys_scale_stride, const float* bias, const float* noise, int noise_batch,
                           const float* noise_w, int act, int B, int C, int H, int W, int in_pitch, unsigned* vmax,
                           int y_fform, int rank_one, void* stream) {
    OODGAN_REQUIRE(z && kernel && y && B > 0 && C > 0 && H > 0 && W > 0, "blur_act_sform: bad args");
    OODGAN_REQUIRE(noise == nullptr || noise_batch == 1 || noise_batch == B, "blur_act_sform: noise_batch");
    OODGAN_REQUIRE(act == OODGAN_ACT_NONE || act == OODGAN_ACT_LRELU, "blur_act_sform: act must be none or lrelu");
    if (in_pitch == 0) in_pitch = 2 * W + 1;
    OODGAN_REQUIRE(in_pitch >= 2 * W + 1 && (in_pitch % 4) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0,
                   "blur_act_sform: the input needs 16-byte aligned rows (pitch %% 4 == 0)");
    BlurArgs a;
    a.z = z; a.kern = kernel; a.y = y; a.ys = reinterpret_cast<uint4*>(ys); a.ys_scale = ys_scale;
    a.ys_scale_stride = ys_scale_stride; a.bias = bias; a.noise = noise; a.noise_w = noise_w; a.noise_batch = noise_batch;
    a.act = act; a.B = B; a.C = C; a.H = H; a.W = W; a.pitch = in_pitch;
    a.tiles_x = (2 * W + 63) / 64;
    a.tiles_y = (2 * H + 7) / 8;
    a.yd = sform_dims(C, 2 * H, 2 * W);
    a.vmax = (ys || y_fform) ? vmax : nullptr;
    a.y_fform = y_fform;
    OODGAN_REQUIRE(!y_fform || (C % 16 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0), "blur_act_fform: C %% 16 == 0 and a 16-byte aligned y");
    // (W even: a thread owns four output columns and masks its noise / plane store per quad — with 2W %% 4 != 0 the last two columns of a row
    // would get neither; odd widths take the tile kernel.  Found by the one-pass up-conv's parity test, round 4.)
    if ((y_fform || ys) && rank_one == 1 && 2 * W >= 64 && (W % 2) == 0 && (C % 16) == 0 && oodgan::tunable(oodgan::OODGAN_TUN_BLUR_STRIP)) {
        BlurStripGeo geo;
        geo.nstrips = (2 * W + 63) / 64;
        const long base = (long)B * a.yd.KC * geo.nstrips;
        long nseg = (8L * 1024) / base;
        if (nseg < 1) nseg = 1;
        int seg_rows = (int)((2 * H + nseg - 1) / nseg);
        if (seg_rows < 32) seg_rows = 32;
        geo.seg_rows = seg_rows;
        geo.nseg = (2 * H + seg_rows - 1) / seg_rows;
        const long nbs = base * geo.nseg;
        OODGAN_REQUIRE(nbs < (1L << 31), "blur_act_fform: grid too large");
        const bool full = ((2 * W) % 64) == 0;
#define OODGAN_BLUR_STRIP_LAUNCH(SF, FULL) \
    hipLaunchKernelGGL((blur_act_fform_strip_kernel<SF, FULL>), dim3((unsigned)nbs), dim3(256), 0, as_stream(stream), a, geo)
        if (y_fform) { if (full) OODGAN_BLUR_STRIP_LAUNCH(false, true); else OODGAN_BLUR_STRIP_LAUNCH(false, false); }
        else { if (full) OODGAN_BLUR_STRIP_LAUNCH(true, true); else OODGAN_BLUR_STRIP_LAUNCH(true, false); }
#undef OODGAN_BLUR_STRIP_LAUNCH
        return check_launch("blur_act_strip");
    }
    const long nb = (long)a.tiles_x * a.tiles_y * a.yd.KC * B;
    OODGAN_REQUIRE(nb < (1L << 31), "blur_act_sform: grid too large");
    hipLaunchKernelGGL(blur_act_sform_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), a);
    return check_launch("blur_act_sform");
}

extern "C" int oodgan_blur_act_sform(const float* z, const float* kernel, float* y, void* ys, const float* ys_scale,
                                     int ys_scale_stride, const float* bias, const float* noise, int noise_batch,
                                     const float* noise_w, int act, int B, int C, int H, int W, int in_pitch, unsigned* vmax,
                                     void* stream) {
    return blur_act_launch(z, kernel, y, ys, ys_scale, ys_scale_stride, bias, noise, noise_batch, noise_w, act, B, C, H, W, in_pitch, vmax, 0, 0, stream);
}

extern "C" int oodgan_blur_act_sform_sep(const float* z, const float* kernel, float* y, void* ys, const float* ys_scale,
                                         int ys_scale_stride, const float* bias, const float* noise, int noise_batch,
                                         const float* noise_w, int act, int B, int C, int H, int W, int in_pitch, unsigned* vmax,
                                         int kernel_rank_one, void* stream) {
    return blur_act_launch(z, kernel, y, ys, ys_scale, ys_scale_stride, bias, noise, noise_batch, noise_w, act, B, C, H, W, in_pitch, vmax, 0,
                           kernel_rank_one, stream);
}

extern "C" int oodgan_blur_act_fform(const float* z, const float* kernel, float* y, const float* ys_scale, int ys_scale_stride,
                                     const float* bias, const float* noise, int noise_batch, const float* noise_w, int act,
                                     int B, int C, int H, int W, int in_pitch, unsigned* vmax, int kernel_rank_one, void* stream) {
    return blur_act_launch(z, kernel, y, nullptr, ys_scale, ys_scale_stride, bias, noise, noise_batch, noise_w, act, B, C, H, W, in_pitch, vmax, 1, kernel_rank_one, stream);
}
