// The 32 -> 32 channel 3x3 convs of the 1024² level inside the W+ loop, reading fp32 activations in F-form
// ([B][2][H][W][16]: one 64-byte record per pixel and 16-channel block) and doing the conversion to the split-f16 operand
// INSIDE the kernel — the passes that used to write and re-read an S-form copy of the same tensor disappear:
//   x_fform = 1 (forward, reference ModulatedConv2d.forward, src/ops/StyleGAN/model.py:233-274): the conv input is the
//       activated output of the up-sampling layer; staged value = x * style[b,k] * range scale, split hi/lo on the fly
//       (was: blur_act_sform writes y AND its S-form, 1.07 GB more per step at B = 8);
//   x_fform = 2 (input gradient): x is the saved OUTPUT `out` of this StyledConv; the kernel runs the layer's activation
//       backward merged with the ToRGB branch (autograd of NoiseInjection / FusedLeakyReLU / ToRGB, model.py:283-292,343-372,
//       src/ops/op/fused_act.py:25-58 — the arithmetic of act_bwd_sform_f_kernel, bwd_producers.hip) on the rows it has just
//       fetched, sums the partial r / t / max of that producer, and feeds g_pre * d * scale to the matrix cores
//       (was: act_bwd_sform_f, 2.45 GB per step, then the S-form read back here).
// Organisation: weights in registers, a workgroup (4 waves) walks down a 32-pixel strip, FOUR rows x 32 pixels per tile (one
// row per wave), bank-conflict-free rotated records as in conv_f16s_strip.hip.  Data path:
//   * EVERY load of the loop is an LDS-DMA (global_load_lds): the raw F-form rows go straight into their place in a ring of
//     five 4-row groups and are converted IN PLACE two tiles later (a pixel's four channel quarters are four adjacent lanes:
//     they read their 16 bytes of the fp32 record, exchange halves by DPP and write the four 16-byte hi / lo slots of the same
//     64 bytes); the halo columns come through a 1 KiB staging area; the per-pixel inputs of the activation backward
//     (g_rgb, noise) and the saved forward input of the style-gradient dot land in LDS the same way.
//     Reason: a VMEM load into registers makes the compiler track vmcnt, and while an LDS-DMA is pending its waitcnt pass
//     turns every such dependency into vmcnt(0) — the round-2 strip kernel drained all memory traffic at the top of every
//     matrix phase that way.  With no register loads the only vmcnt waits are the explicit counted ones below, which leave two
//     tiles of prefetch and the newest stores in flight (memory operations of a wave retire in issue order);
//   * the operation counts per tile and wave are uniform (clamped duplicates instead of branches; H % 4 == 0, W % 32 == 0):
//     backward 12 DMA + 16 stores, forward 6 DMA + 4 (+3) stores — far below the 64 the 6-bit counter allows.
#include "conv_common.hpp"
#include "sform.hpp"
#include <cstdint>
#include <cstdlib>

using namespace oodgan;

#include "stripx_common.hpp"

#ifdef OODGAN_CLOCK_STAMP
__device__ unsigned long long* g_stripx_stamp = nullptr;
__device__ long g_stripx_stamp_n = 0;
#endif

namespace {

// BWD: x_fform == 2 (activation backward in the conversion, style-gradient dot in the epilogue).  RGB: fused ToRGB colour
// sums (forward).  PRE: y <- dx * act'(dotx) (oodgan_conv_args.dot_actgrad).  YF: y in F-form.  SEG: the strips are cut into
// segments (fewer strips than CUs): rows fetched as another segment's halo must be kept out of the backward's sums; with one
// segment per strip the only such rows lie outside the image and contribute zeros by themselves.
// G2 (input-gradient instances with oodgan_conv_args.x_hi_only, precision 'f16s-g2'): the converted operand keeps its hi half only —
// g_hi * (w_hi + w_lo): six matrix instructions per chunk instead of nine, three fragment reads instead of six, no lo split and half
// the LDS writes of the in-place conversion
template <bool BWD, bool RGB, bool PRE, bool YF, bool SEG = true, bool G2 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_f16s_stripx_kernel(
    const StripX p, const uint4* __restrict__ wpk16) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using L = SXL<BWD>;
    constexpr int SX_NG = L::NG, NB = L::NB, PD = L::PD;
    constexpr int SX_HALO = L::HALO, SX_SMALL = L::SMALL, SX_DOT = L::DOT, SX_FIN = L::FIN, SX_CST = L::CST, SX_EPC = L::EPC;
    const oodgan_conv_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    int w = blockIdx.x;
    {
        const int total = gridDim.x, xcd = w & 7, idx = w >> 3, q = total >> 3, r = total & 7;
        w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = w % p.tiles_x;
    const int seg = (w / p.tiles_x) % p.nseg;
    const int b = w / (p.tiles_x * p.nseg);
    const int t0 = seg * p.seg_tiles;
    const int n = min(p.seg_tiles, p.tiles_y - t0);
    const int c0 = tx * 32, R0 = 4 * t0;
    const int H = a.Hin, W = a.Win;
    constexpr int M = 32;
    const long HW = (long)H * W;

    // ---- weights: the whole tensor in registers.  Packed order (oodgan_pack_conv3x3_f16s): [kc][tap][hi|lo][k-half][Mp][8 f16]
    half8 ah[9][2], al[9][2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const half8* wb = reinterpret_cast<const half8*>(wpk16) + ((long)(kc * 9 + tp) * 4) * p.Mp;
            ah[tp][kc] = wb[(0 * 2 + half) * p.Mp + l31];
            al[tp][kc] = wb[(1 * 2 + half) * p.Mp + l31];
        }
#if 1
    // home the weight fragments in accumulation registers: v_mfma reads its A operand from either file, the VALU work of the loop
    // then has the architectural registers to itself (otherwise the allocator parks them there as spills and copies them back
    // before every use: 76-204 v_accvgpr_read per tile)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            asm volatile("" : "+a"(ah[tp][kc]));
            asm volatile("" : "+a"(al[tp][kc]));
        }
#endif

    // ---- per-workgroup constants in LDS (read back where they are used: they would otherwise occupy registers through the
    // matrix phase): epilogue scale out_scale * unscale of channel m(r) = (r & 3) + 8 (r >> 2) + 4 half; the seven per-channel
    // constants of the activation backward for every (channel block, quarter)
    const float us = (p.w_unscale ? p.w_unscale[0] : 1.f) * (a.in_mul2 ? a.in_mul2[0] : 1.f);
    if (tid < 32) {
        const int hh = tid >> 4, r = tid & 15, m = (r & 3) + 8 * (r >> 2) + 4 * hh;
        reinterpret_cast<float*>(smem + SX_CST)[tid] = (a.out_scale ? a.out_scale[(long)b * a.out_scale_stride + m] : 1.f) * us;
    }
    float4* cst = reinterpret_cast<float4*>(smem + SX_CST + 128);      // [kc*4 + q][w0, w1, w2 (x rgb_scale), s_rgb*sqrt2, s_rgb*0.2*sqrt2, bias, d*scale]
    float nwb = 0.f;
    if (BWD) {
        const oodgan_actbwd_fuse& f = p.f;
        nwb = f.noise ? (f.noise_w ? f.noise_w[0] : 1.f) : 0.f;
        if (tid < 8) {
            const float m2 = f.mul2 ? f.mul2[1] : 1.f;
            float c7[7][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = tid * 4 + j;
                c7[0][j] = f.w_rgb[0 * M + c] * f.rgb_scale;
                c7[1][j] = f.w_rgb[1 * M + c] * f.rgb_scale;
                c7[2][j] = f.w_rgb[2 * M + c] * f.rgb_scale;
                const float srv = f.s_rgb[(long)b * f.s_rgb_stride + c];
                c7[3][j] = srv * kSqrt2;                       // s_rgb folded into the two slopes of the activation gradient
                c7[4][j] = srv * (0.2f * kSqrt2);
                c7[5][j] = f.bias ? f.bias[c] : 0.f;
                const float dv = f.dscale[(long)b * f.dscale_stride + c];
                c7[6][j] = dv * m2;
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) cst[tid * 7 + k] = make_float4(c7[k][0], c7[k][1], c7[k][2], c7[k][3]);
        }
    }
    if (!BWD && tid < 32) {      // forward epilogue constants of channel m(r): bias and the three modulated ToRGB rows (rgb_scale * w[k,m] * s_rgb[b,m])
        const int hh = tid >> 4, r = tid & 15, m = (r & 3) + 8 * (r >> 2) + 4 * hh;
        float* e = reinterpret_cast<float*>(smem + SX_EPC) + (hh * 4 + (r >> 2)) * 16 + (r & 3);
        e[0] = a.bias ? a.bias[m] : 0.f;
        const float sv = RGB ? a.rgb_scale * a.rgb_s[(long)b * a.rgb_s_stride + m] : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) e[4 * (k + 1)] = RGB ? sv * a.rgb_w[k * M + m] : 0.f;
    }
    const float nwf = (!BWD && a.noise) ? (a.noise_w ? a.noise_w[0] : 1.f) : 0.f;
    unsigned moff[YF ? 1 : 4];               // NCHW y: byte offset of channel 8 rr + 4 half at this lane's column
    if (!YF) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) moff[rr] = (unsigned)((long)(8 * rr + 4 * half) * p.out_plane * 4) + l31 * 4;
    }

    // ---- conversion role of this thread: channel block ckc = wave >> 1, pixel cpx = 16 (wave & 1) + (lane >> 2) of the strip,
    // channel quarter cq = lane & 3 (channels ckc*16 + cq*4 + 0..3), in each of the group's four rows: a wave's DMA piece for
    // one row is 16 records = 1 KiB contiguous in the ring.  Halo columns 0 / 33 of row `wave`: lanes 0..15 = (side, kc, quarter).
    // Rows / columns outside the image read a page of zeros (zero padding of the conv; zero `out` and zero g_rgb give a zero
    // gradient and zero sums): no per-unit masks.
    const int ckc = wave >> 1, cpx = 16 * (wave & 1) + (lane >> 2), cq = lane & 3;
    const bool even = (cq & 1) == 0;
    const int slot = even ? (cq >> 1) : 2 + (cq >> 1);
    const unsigned crec = (unsigned)(ckc * (SX_C * 64) + (1 + cpx) * 64);                      // record of the pixel inside a ring row
    const unsigned cwr = crec + ((((unsigned)slot + (((1 + cpx) >> 2) & 3)) & 3) << 4);          // the slot this lane writes (rotated)
    const unsigned goff0 = (unsigned)((((long)ckc * HW + c0 + cpx) * 16 + cq * 4) * 4);
    const int hside = (lane >> 3) & 1, hkc = (lane >> 2) & 1, hcol = hside ? 33 : 0, hgx = c0 - 1 + hcol;
    const bool hinv = hgx < 0 || hgx >= W;
    const unsigned hgoff = (unsigned)((((long)hkc * HW + min(max(hgx, 0), W - 1)) * 16 + cq * 4) * 4);
    const unsigned hwr = (unsigned)(hkc * (SX_C * 64) + hcol * 64 + ((((unsigned)slot + ((hcol >> 2) & 3)) & 3) << 4));
    const unsigned char* xfb = reinterpret_cast<const unsigned char*>(a.x) + (long)b * 2 * HW * 64;
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(p.zeros);
    const long row_bytes = (long)W * 64;

    float xs[4], xsh[4];                     // forward: staged value = x * in_scale[b,k] * in_mul2[1] (interior / halo role)
    float acc_r[BWD ? 4 : 1], acc_t[BWD ? 4 : 1], amax4[BWD ? 4 : 1];
    if (!BWD) {
        const float m2 = a.in_mul2 ? a.in_mul2[1] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xs[j] = (a.in_scale ? a.in_scale[(long)b * a.in_scale_stride + ckc * 16 + cq * 4 + j] : 1.f) * m2;
            xsh[j] = (a.in_scale ? a.in_scale[(long)b * a.in_scale_stride + hkc * 16 + cq * 4 + j] : 1.f) * m2;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_r[j] = acc_t[j] = amax4[j] = 0.f;
    }

    // backward: component lane & 3 of the per-pixel inputs (g_rgb 0..2, noise) of pixel slot (lane >> 2) of a piece
    const float* sbase = nullptr;
    bool nz_zero = false;
    if (BWD) {
        const oodgan_actbwd_fuse& f = p.f;
        const int comp = lane & 3;
        sbase = f.g_rgb + ((long)b * 3 + (comp < 3 ? comp : 0)) * HW;
        if (comp == 3) {
            nz_zero = f.noise == nullptr;
            if (f.noise) sbase = f.noise + (long)(f.noise_batch > 1 ? b : 0) * HW;
        }
    }
    const float* nzb = (!BWD && a.noise) ? a.noise + (long)(a.noise_batch > 1 ? b : 0) * HW : nullptr;
    const unsigned char* dfb = BWD ? reinterpret_cast<const unsigned char*>(a.dotx) + ((long)b * 2 * HW) * 64 + half * 16 : nullptr;

    // ---- one batch of LDS-DMA: raw rows + halo + small inputs of group g (into ring group gs, buffers g3) and, with it, the
    // epilogue inputs of tile td (dot rows / noise; buffer d3).  Groups past the image read the page of zeros; the caller clamps
    // td to a valid tile (the duplicates land in dead buffers).  12 operations per wave (backward), 6 (forward).
    // The kBatch operations of a batch, one at a time (`op` is a compile-time index after unrolling): the loop issues them spread
    // over the chunks of the matrix phase.  Issued all at once at the top of the tile the ~140 cache lines a wave requests do not
    // fit the memory pipeline's queues — the wave sat in the issue of its loads for as long as HBM took to serve them (forward:
    // 793 us, 552 us without the loads; the difference is the 1.07 GB at the HBM rate), every CU at the same moment.
    auto issue_op = [&](int op, int g, int gs, int g3, int td, int d3) {
        const int r0g = R0 + 4 * g + 1;
        if (op < 4) {                                // raw row `op` of the thread's pixel
            const int r = r0g + op;
            const bool rok = r >= 0 && r < H;                        // wave-uniform
            const unsigned char* src = rok ? xfb + (long)r * row_bytes + goff0 : zp;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + gs * SX_GROUP + op * SX_ROW + ckc * (SX_C * 64) + (1 + 16 * (wave & 1)) * 64), 16, 0, SX_NT_LD);
        } else if (op == 4) {                        // halo records of row `wave`
            const int r = r0g + wave;
            const unsigned char* src = (hinv || r < 0 || r >= H) ? zp : xfb + (long)r * row_bytes + hgoff;
            if (lane < 16)
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + SX_HALO + g3 * 1024 + wave * 256), 16, 0, SX_NT_LD);
        } else if (!BWD) {                           // op 5: the noise of tile td
            const float* src = nzb ? nzb + (long)(R0 + 4 * td + wave) * W + c0 + l31 : reinterpret_cast<const float*>(zp);
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + SX_SMALL + d3 * 1024 + wave * 256), 4, 0, 0);
        } else if (op < 7) {                         // per-pixel inputs, pieces 0..7: row q >> 1, pixels 16 (q & 1) .. + 15
            const int q = wave + 4 * (op - 5);
            const int r = r0g + (q >> 1);
            const bool rok = r >= 0 && r < H;
            const float* src = (rok && !nz_zero) ? sbase + (long)r * W + c0 + 16 * (q & 1) + (lane >> 2) : reinterpret_cast<const float*>(zp);
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + SX_SMALL + g3 * SX_SMALL_ONE + q * 256), 4, 0, 0);
        } else if (op == 7) {                        // piece 8 (+ duplicates 9..11): halo pixels, slot 128 + row*2 + side
            const int hs = (lane >> 2) & 7;
            const int r = r0g + (hs >> 1), gx = c0 - 1 + ((hs & 1) ? 33 : 0);
            const bool ok = r >= 0 && r < H && gx >= 0 && gx < W && !nz_zero;
            const float* src = ok ? sbase + (long)r * W + gx : reinterpret_cast<const float*>(zp);
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + SX_SMALL + g3 * SX_SMALL_ONE + (8 + wave) * 256), 4, 0, 0);
        } else {                                     // ops 8..11: the saved forward input of tile td (style-gradient dot)
            const int rr = op - 8;
            const long pix = (long)(R0 + 4 * td + wave) * W + c0 + l31;
            __builtin_amdgcn_global_load_lds((gbl_void*)(dfb + ((long)(rr >> 1) * HW + pix) * 64 + (rr & 1) * 32),
                                             (lds_void*)(smem + SX_DOT + d3 * SX_DOT_ONE + wave * 4096 + rr * 1024), 16, 0, SX_NT_LD);
        }
    };
    auto issue_batch = [&](int g, int gs, int g3, int td, int d3) {
#pragma unroll
        for (int op = 0; op < (BWD ? 12 : 6); ++op) issue_op(op, g, gs, g3, td, d3);
    };

    // ---- in-place conversion of group g (ring group gs, buffers g3): fp32 records -> rotated hi / lo slots.  `count`: the
    // group's rows enter this workgroup's sums (wave-uniform; the caller excludes duplicates).
    auto convert_unit = [&](const f32x4 rv, const f32x4 s4, const f32x4 (&cq7)[7], const float* xsc, float fi, unsigned dst, bool wr_ok, bool sums) {
        const float ov[4] = {rv[0], rv[1], rv[2], rv[3]};
        float v[4];
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ov[j] * xsc[j];
        } else {
            const f32x4 w0 = cq7[0], w1 = cq7[1], w2 = cq7[2], sa = cq7[3], sb = cq7[4], bv = cq7[5], ds = cq7[6];
            const float nz = nwb * s4[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float o = ov[j];
                const float t = w0[j] * s4[0] + w1[j] * s4[1] + w2[j] * s4[2];
                const bool pos = o > 0.f;
                const float gp = t * (pos ? sa[j] : sb[j]);
                const float ycv = o * (pos ? kInvPos : kInvNeg) - (nz + bv[j]);
                if (sums) {          // false for the halo unit: its pixels are counted by the neighbouring strip
                    const float gi = SEG ? gp * fi : gp;
                    acc_r[BWD ? j : 0] += gi * ycv;
                    acc_t[BWD ? j : 0] += (SEG ? o * fi : o) * t;
                    amax4[BWD ? j : 0] = fmaxf(amax4[BWD ? j : 0], fabsf(gi));     // x |d| at the end
                }
                v[j] = gp * ds[j];
            }
        }
        if (G2) {
            // hi halves only: the odd quarter hands its two packed pairs to the even one, which writes the record's hi slot
            oodgan_half2v p01, p23;
            p01[0] = (_Float16)v[0]; p01[1] = (_Float16)v[1];
            p23[0] = (_Float16)v[2]; p23[1] = (_Float16)v[3];
            const unsigned h01 = __builtin_bit_cast(unsigned, p01), h23 = __builtin_bit_cast(unsigned, p23);
            const unsigned g0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)h01, 0xB1, 0xF, 0xF, false);
            const unsigned g1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)h23, 0xB1, 0xF, 0xF, false);
            if (wr_ok && even) lds_write16(dst, h01, h23, g0, g1);
            return;
        }
        unsigned h01, l01, h23, l23;
        split_pair(v[0], v[1], h01, l01);
        split_pair(v[2], v[3], h23, l23);
        // quarters (0,1) and (2,3) exchange: the even one collects the hi halves of the 8 channels, the odd one the lo halves
        const unsigned s0 = even ? l01 : h01, s1 = even ? l23 : h23;
        const unsigned g0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
        const unsigned g1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, false);
        if (wr_ok) lds_write16(dst, even ? h01 : g0, even ? h23 : g1, even ? g0 : l01, even ? g1 : l23);
    };
    // interior role: the seven constant vectors stay in registers; the halo role (lanes 0..15, another channel block) reads its own
    f32x4 cin[7];
    if (BWD) {
        __syncthreads();                             // the constant table is complete
        lds_read7_16(lds_addr(cst + (ckc * 4 + cq) * 7), cin);
    }
    const unsigned smem0 = lds_addr(smem);
    auto convert = [&](int g, int gs, int g3, bool count) {
        const unsigned ring = smem0 + gs * SX_GROUP;
        f32x4 rv[5], s4[5];
        // all four quarters of a record have read it before any of them writes (one instruction stream per wave; the block
        // ends with lgkmcnt(0))
        lds_read_rows(smem0 + gs * SX_GROUP + crec + cq * 16, smem0 + SX_HALO + g3 * 1024 + wave * 256 + (lane & 15) * 16, rv[0], rv[1], rv[2], rv[3], rv[4]);
        f32x4 chl[7];
        if (BWD) {
            const unsigned sm = smem0 + SX_SMALL + g3 * SX_SMALL_ONE;
            lds_read_small(sm + cpx * 16, sm + (128 + wave * 2 + hside) * 16, s4[0], s4[1], s4[2], s4[3], s4[4]);
            lds_read7_16(smem0 + SX_CST + 128 + (hkc * 4 + cq) * 7 * 16, chl);
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i) s4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 7; ++i) chl[i] = cin[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // rows counted in this workgroup's sums: its own 4n rows (group -1 contributes its last row, the last group all
            // but its last); the halo columns belong to the neighbouring strips
            const float fi = (count && (g < 0 ? i == 3 : (g < n - 1 || i < 3))) ? 1.f : 0.f;
            convert_unit(rv[i], s4[i], cin, xs, fi, ring + i * SX_ROW + cwr, true, true);
        }
        convert_unit(rv[4], s4[4], chl, xsh, 0.f, ring + wave * SX_ROW + hwr, lane < 16, false);
    };

    __builtin_amdgcn_s_waitcnt(SX_VM(0));            // weights, scales: retired here, never inside the loop
    // ---- prologue: groups -1 (its last two rows are the halo above the first tile) and 0 converted, then the batches "-PD" .. "-1":
    // groups 1 .. PD in flight, with them the epilogue inputs of the tiles up to PD - 2
    issue_batch(-1, SX_NG - 1, NB - 1, 0, 0);
    issue_batch(0, 0, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(SX_VM(0));
    __syncthreads();
    convert(-1, SX_NG - 1, NB - 1, true);
    convert(0, 0, 0, true);
    __syncthreads();                                 // every wave is done with the small-input buffers before later groups land in them
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = PD; k >= 1; --k) {                  // batch(-k): group PD + 1 - k, epilogue inputs of tile PD - 1 - k (k = PD: never read)
        const int g = PD + 1 - k, td = PD - 1 - k;
        issue_batch(g, g % SX_NG, g % NB, min(max(td, 0), n - 1), (td + NB) % NB);
    }
    __builtin_amdgcn_sched_barrier(0);

    // lane-constant part of the fragment addresses: record kx + l31, slot (half + 2*lo) rotated by (c>>2)&3
    unsigned lrd[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
            const int c = kx + l31;
            lrd[kx][lo] = c * 64 + (((half + 2 * lo + ((c >> 2) & 3)) & 3) << 4);
        }
    const int px = c0 + l31;
    constexpr int kBatch = BWD ? 12 : 6;
    constexpr int kStores = (YF ? 4 : 16) + (RGB ? 3 : 0);
    constexpr int kTail = (YF ? 1 : 4) + (RGB ? 3 : 0);         // stores issued after the last operation of the iteration's batch
    int rb = 4 * SX_NG - 2 + wave;   // ring row of image row R0 + 4t - 1 + wave (group g, row j at ring row 4 (g mod NG) + j): group -1 is the last ring group
    int gs1 = 1;             // ring group of group t + 1
    int m3 = 0;              // t mod NB
    int m3l = NB - 1;        // (t - 1) mod NB
    float vprev[16], dsum[BWD ? 16 : 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) vprev[r] = 0.f;
    if (BWD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dsum[BWD ? r : 0] = 0.f;
    }

    // Iteration t: matrix phase of tile t with the in-place conversion of group t+1 AND the epilogue arithmetic of tile t-1 woven in
    // (a wave64 VALU instruction takes 4 cycles, a matrix instruction keeps its pipe busy for 32: the ~500 VALU instructions of a
    // tile fit under its 54 matrix instructions only if they sit between them); then the stores of tile t-1.  One more
    // iteration than tiles; the matrix phase of iteration n works on dead data.
#ifdef OODGAN_CLOCK_STAMP
    unsigned long long st_wait = 0, st_bar = 0, st_mat = 0, st_sto = 0;
    const unsigned long long st_l0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = 0; t <= n; ++t) {
        SX_STAMP(st_a);
        // Issue order of an iteration: the batch's operations chunk by chunk, the y stores of a slice after chunks 1, 2, 3 and 5, the
        // colour stores at the end — so behind the LAST operation of batch(t-PD) come kTail stores of its own iteration and PD - 1
        // whole iterations (kBatch + kStores each).  Everything up to that operation must have landed.  For t < PD the batch in
        // question was issued before the loop: behind it come PD - 1 - t more of those and the t iterations so far (iteration 0 stores
        // like any other — into tile 0's rows, see below).
        if (t >= PD) __builtin_amdgcn_s_waitcnt(SX_VML(kTail + (PD - 1) * (kBatch + kStores)));
        else if (t == 3) __builtin_amdgcn_s_waitcnt(SX_VML((PD - 1) * kBatch + 3 * kStores));
        else if (t == 2) __builtin_amdgcn_s_waitcnt(SX_VML((PD - 1) * kBatch + 2 * kStores));
        else if (t == 1) __builtin_amdgcn_s_waitcnt(SX_VML((PD - 1) * kBatch + kStores));
        else __builtin_amdgcn_s_waitcnt(SX_VML((PD - 1) * kBatch));
        static_assert(PD <= 4 && kTail + (PD - 1) * (kBatch + kStores) < 64, "the counted wait fits vmcnt");
        SX_STAMP(st_b);
        __builtin_amdgcn_s_barrier();                // group t is converted; ring group (t+PD+1) % NG and the (t % NB) buffers are free
        SX_STAMP(st_c);
        __builtin_amdgcn_sched_barrier(0);
        const int m3n = m3 == NB - 1 ? 0 : m3 + 1;   // (t + 1) % NB
        int gsn = gs1 + PD;                          // ring group of group t + PD + 1
        if (gsn >= SX_NG) gsn -= SX_NG;
        int md = m3 - 2;                             // buffer of tile t + PD - 1
        if (md < 0) md += NB;
        const int tdn = min(t + PD - 1, n - 1);
        // batch(t) — group t + PD + 1 (past the segment's end: rows of the next segment, kept out of the sums by `cnt_ok`, or rows
        // below the image, the page of zeros) and the epilogue inputs of tile t + PD - 1 — is issued in kBatch / 6 operations per chunk
        // The compiler's waitcnt model counts an LDS-DMA as an outstanding LDS access too and, while one is pending, turns every
        // lgkmcnt wait into lgkmcnt(0) — operand fragments requested a chunk ahead would be drained at every use.  The hardware
        // counts the DMA in vmcnt only: this wait costs nothing (no LDS operation is outstanding here) and retires the DMAs in
        // the model, so that the waits below are counted.
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);

        const bool cnt_ok = t + 1 < n;
        const unsigned cring = smem0 + gs1 * SX_GROUP;
        const unsigned csm = smem0 + SX_SMALL + m3n * SX_SMALL_ONE;
        const float eflag = t >= 1 ? us : 0.f;       // the epilogue of "tile -1" contributes nothing to the sums
        // row of tile t-1 this wave stores; iteration 0 has nothing to store and writes (garbage) to tile 0's row, which iteration 1
        // overwrites — same wave, same addresses, in order — instead of branching around the stores inside the matrix phase
        const int pyp = R0 + 4 * max(t - 1, 0) + wave;
        float* const yfp = a.y + (((long)b * 2 * H + pyp) * W + px) * 16 + 4 * half;                                   // F-form
        unsigned char* const yrp = reinterpret_cast<unsigned char*>(a.y) + ((long)b * M * p.out_plane + (long)pyp * a.out_pitch + c0) * 4;   // NCHW
        unsigned rbase[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int row = rb + ky;
            if (row >= 4 * SX_NG) row -= 4 * SX_NG;
            rbase[ky] = smem0 + row * SX_ROW;
        }
        // Every LDS read of the tile is issued one chunk ahead of its use (chunk = the nine matrix instructions of one channel
        // block and tap row) and waited for at the end of the chunk in between: operand fragments of chunk c+1, the raw record (and
        // per-pixel inputs) of conversion unit c+1, the constants of the next epilogue slice.
        half8 fb[2][6];
        f32x4 ru[2], su[2], chl[7], osc[4], dxv[4], ec4[2][4];
        float nzr = 0.f;
        auto frag_issue = [&](int c, half8 (&f)[6]) {
            const unsigned base = rbase[c % 3] + (c / 3) * (SX_C * 64);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                lds_issue(f[kx], base + lrd[kx][0]);
                if (!G2) lds_issue(f[3 + kx], base + lrd[kx][1]);
            }
        };
        auto unit_issue = [&](int u, f32x4& r, f32x4& s4) {       // u = 0..3: row u of the thread's pixel; 4: the halo record
            if (u < 4) {
                lds_issue(r, cring + u * SX_ROW + crec + cq * 16);
                if (BWD) lds_issue(s4, csm + (u * 32 + cpx) * 16);
            } else {
                lds_issue(r, smem0 + SX_HALO + m3n * 1024 + wave * 256 + (lane & 15) * 16);
                if (BWD) lds_issue(s4, csm + (128 + wave * 2 + hside) * 16);
            }
            if (!BWD) s4 = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        auto slice_of = [](int c) { return c == 1 ? 0 : c == 2 ? 1 : c == 3 ? 2 : c == 5 ? 3 : -1; };
        // first batch of the tile (its latency is the one exposed per tile): fragments of chunk 0, unit 0, the epilogue inputs of tile t-1
        frag_issue(0, fb[0]);
        unit_issue(0, ru[0], su[0]);
#pragma unroll
        for (int k = 0; k < 4; ++k) lds_issue(osc[k], smem0 + SX_CST + half * 64 + k * 16);
        if (BWD) {
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_issue(dxv[k], smem0 + SX_DOT + m3l * SX_DOT_ONE + wave * 4096 + lane * 16 + k * 1024);
        } else {
            lds_issue(nzr, smem0 + SX_SMALL + m3l * 1024 + wave * 256 + lane * 4);
        }
        lds_wait(fb[0]);
        lds_wait(ru[0], su[0]);
        lds_wait(osc[0], osc[1], osc[2], osc[3]);
        if (BWD) lds_wait(dxv[0], dxv[1], dxv[2], dxv[3]);
        else lds_wait(nzr);
        const float nz = nwf * nzr;
        float o[16];
        float c0s = 0.f, c1s = 0.f, c2s = 0.f;

        // two accumulation chains: hi*hi, and the two cross terms hi*lo + lo*hi (the small terms are summed among themselves before
        // they meet the large one).  Three chains were slower (16 more registers): the kernel is bound by the ISSUE of its ~700
        // instructions per tile and wave — one wave per SIMD issues one instruction every 4-5 cycles whatever its kind.
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int c = 0; c < ((SX_ABL & 4) ? 0 : 6); ++c) {
            const int kc = c / 3, ky = c % 3;
            // ---- this chunk's share of batch(t), then the LDS reads for chunk c+1
            if (!(SX_ABL & 8)) {
#pragma unroll
                for (int k = 0; k < kBatch / 6; ++k) issue_op(c * (kBatch / 6) + k, t + PD + 1, gsn, m3, tdn, md);
            }
            if (c < 5) frag_issue(c + 1, fb[(c + 1) & 1]);
            if (c < 4) unit_issue(c + 1, ru[(c + 1) & 1], su[(c + 1) & 1]);
            if (BWD && c == 3) {
#pragma unroll
                for (int k = 0; k < 7; ++k) lds_issue(chl[k], smem0 + SX_CST + 128 + ((hkc * 4 + cq) * 7 + k) * 16);
            }
            const int ecn = c < 5 ? slice_of(c + 1) : -1;
            if (!BWD && ecn >= 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) lds_issue(ec4[(c + 1) & 1][k], smem0 + SX_EPC + (half * 4 + ecn) * 64 + k * 16);
            }
            // ---- chunk c: nine matrix instructions ...
            const half8 (&f)[6] = fb[c & 1];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tp = ky * 3 + kx;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tp][kc], f[kx], acc0, 0, 0, 0);
                if (!G2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tp][kc], f[3 + kx], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ky * 3 + kx][kc], f[kx], acc1, 0, 0, 0);
            // ---- ... with conversion unit c (in place: the four quarters of a record have all read it — their reads were waited for
            // at the end of the previous chunk — before any of them writes; past the segment's end a dead ring group, not counted)
            if (!(SX_ABL & 2)) {
                if (c < 4) {
                    // rows counted in this workgroup's sums: its own 4n rows (the last group all but its last row); the halo
                    // columns belong to the neighbouring strips
                    const float fi = (cnt_ok && (t + 1 < n - 1 || c < 3)) ? 1.f : 0.f;
                    convert_unit(ru[c & 1], su[c & 1], cin, xs, fi, cring + c * SX_ROW + cwr, true, true);
                } else if (c == 4) {
                    convert_unit(ru[0], su[0], chl, xsh, 0.f, cring + wave * SX_ROW + hwr, lane < 16, false);
                }
            }
            // ---- ... and a slice of the epilogue arithmetic of tile t-1: channels 4ec .. 4ec+3 after chunks 1, 2, 3, 5
            const int ec = slice_of(c);
            if (ec >= 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = ec * 4 + e;
                    const float v = vprev[r];
                    float ov = v * osc[ec][e];
                    if (BWD) {
                        const float dv = dxv[ec][e];
                        dsum[BWD ? r : 0] += (v * eflag) * dv;
                        if (PRE) ov *= dv > 0.f ? kSqrt2 : 0.2f * kSqrt2;
                    } else {
                        ov += nz + ec4[c & 1][0][e];
                        if (a.act == OODGAN_ACT_LRELU) ov = (ov > 0.f ? ov : 0.2f * ov) * kSqrt2;
                    }
                    o[r] = ov;
                    if (RGB) {
                        c0s += ec4[c & 1][1][e] * ov;
                        c1s += ec4[c & 1][2][e] * ov;
                        c2s += ec4[c & 1][3][e] * ov;
                    }
                }
                // the slice's four channels leave at once (spread over the tile like the loads: a burst of stores at the end of
                // the tile stalled in its issue — 1.0 of 3.6 us per tile in the stamp build)
                if (!(SX_ABL & 1)) {
                    if (YF) {
                        *reinterpret_cast<float4*>(yfp + (long)(ec >> 1) * HW * 16 + (ec & 1) * 8) = make_float4(o[4 * ec], o[4 * ec + 1], o[4 * ec + 2], o[4 * ec + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            *reinterpret_cast<float*>(yrp + (long)e * p.out_plane * 4 + moff[YF ? 0 : ec]) = o[4 * ec + e];
                    }
                }
            }
            // ---- what was issued at the top of this chunk has had nine matrix instructions to arrive
            if (c < 5) lds_wait(fb[(c + 1) & 1]);
            if (c < 4) lds_wait(ru[(c + 1) & 1], su[(c + 1) & 1]);
            if (BWD && c == 3) lds_wait(chl);
            if (!BWD && ecn >= 0) lds_wait(ec4[(c + 1) & 1][0], ec4[(c + 1) & 1][1], ec4[(c + 1) & 1][2], ec4[(c + 1) & 1][3]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) vprev[r] = acc0[r] + acc1[r];
        __builtin_amdgcn_sched_barrier(0);
        SX_STAMP(st_d);
        // ---- the colour sums of tile t-1 (its y stores went out slice by slice above)
        if (RGB) {
            // the lane's 16 channels of the three colour sums; the other 16 channels sit in lane ^ 32
            c0s += __shfl_xor(c0s, 32, 64);
            c1s += __shfl_xor(c1s, 32, 64);
            c2s += __shfl_xor(c2s, 32, 64);
            if (half == 0) {
                float* rp = a.rgb_y + (long)b * 3 * HW + (long)pyp * W + px;
                rp[0] = c0s;
                rp[HW] = c1s;
                rp[2 * HW] = c2s;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        rb += 4;
        if (rb >= 4 * SX_NG) rb -= 4 * SX_NG;
        gs1 = gs1 == SX_NG - 1 ? 0 : gs1 + 1;
        m3l = m3;
        m3 = m3n;
#ifdef OODGAN_CLOCK_STAMP
        {
            const unsigned long long st_e = __builtin_amdgcn_s_memtime();
            st_wait += st_b - st_a; st_bar += st_c - st_b; st_mat += st_d - st_c; st_sto += st_e - st_d;
        }
#endif
    }
#ifdef OODGAN_CLOCK_STAMP
    if (lane == 0 && g_stripx_stamp && (long)blockIdx.x < g_stripx_stamp_n) {
        unsigned long long* q = g_stripx_stamp + ((long)blockIdx.x * 4 + wave) * 6;
        q[0] = st_wait; q[1] = st_bar; q[2] = st_mat; q[3] = st_sto;
        q[4] = __builtin_amdgcn_s_memtime() - st_l0; q[5] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
    __builtin_amdgcn_s_waitcnt(SX_VML(0));
    __syncthreads();
    if (BWD) {
        const oodgan_actbwd_fuse& f = p.f;
        const int part = seg * p.tiles_x + tx;
        float* fin = reinterpret_cast<float*>(smem + SX_FIN);         // [wave][32] dot sums, then [wave][quarter*4 + j][r, t]
        float* finm = fin + 4 * 32 * 2;
        // style-gradient dot: the 32 pixels of each half wave, then the four waves (rows)
#pragma unroll
        for (int r = 0; r < 16; ++r) dsum[BWD ? r : 0] = half_sum_dpp(dsum[BWD ? r : 0]);
        if (l31 == kHalfSumLane) {
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = dsum[BWD ? r : 0];
        }
        __syncthreads();
        if (tid < 32) a.dot_part[((long)b * M + tid) * a.dot_nparts + part] = (fin[tid] + fin[32 + tid]) + (fin[64 + tid] + fin[96 + tid]);
        __syncthreads();
        // partial sums of the activation backward: the 16 lanes of a wave with the same quarter (lane bits 2-5); the waves
        // (ckc = wave >> 1, two waves per channel block) are combined by the first 32 threads
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o_ = 4; o_ < 64; o_ <<= 1) {
                acc_r[BWD ? j : 0] += __shfl_xor(acc_r[BWD ? j : 0], o_, 64);
                acc_t[BWD ? j : 0] += __shfl_xor(acc_t[BWD ? j : 0], o_, 64);
            }
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, amax4[BWD ? j : 0] * fabsf(f.dscale[(long)b * f.dscale_stride + ckc * 16 + cq * 4 + j]));
#pragma unroll
        for (int o_ = 1; o_ < 64; o_ <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o_, 64));
        if (lane < 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fin[(wave * 16 + lane * 4 + j) * 2 + 0] = acc_r[BWD ? j : 0];
                fin[(wave * 16 + lane * 4 + j) * 2 + 1] = acc_t[BWD ? j : 0];
            }
        }
        if (lane == 0) finm[wave] = amax;
        __syncthreads();
        if (tid < 32) {      // channel tid: block tid >> 4 = waves 2 (tid >> 4) and + 1
            const int wv = 2 * (tid >> 4), cc = tid & 15;
            const long o_ = ((long)b * M + tid) * p.nparts + part;
            f.part_r[o_] = fin[(wv * 16 + cc) * 2] + fin[((wv + 1) * 16 + cc) * 2];
            f.part_t[o_] = fin[(wv * 16 + cc) * 2 + 1] + fin[((wv + 1) * 16 + cc) * 2 + 1];
        }
        if (tid < 2) f.part_max[((long)b * 2 + tid) * p.nparts + part] = fmaxf(finm[2 * tid], finm[2 * tid + 1]);
    }
}

const void* zero_page() {
    static void* const z = [] {          // initialised once, thread-safe (function-local static)
        void* q = nullptr;
        if (hipMalloc(&q, 256) != hipSuccess || hipMemset(q, 0, 256) != hipSuccess) q = nullptr;
        return q;
    }();
    return z;
}

int num_cus() {
    static const int num_cu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        return (int)prop.multiProcessorCount;
    }();
    return num_cu;
}

// everything that must not happen for the first time inside a stream capture (hipMalloc of the zero page, device query, function
// attributes): done on the first oodgan_conv3x3_xf_supported() call — the engine makes one when it is built — or launch
bool stripx_init() {
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<true, false, false, false, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<true>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<true, false, true, false, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<true>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<true, false, false, false, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<true>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<true, false, true, false, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<true>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<true, false, false, false, true, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<true>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<true, false, true, false, true, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<true>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<true, false, false, false, false, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<true>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<true, false, true, false, false, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<true>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<false, false, false, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<false>::SMEM),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f16s_stripx_kernel<false, true, false, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, SXL<false>::SMEM), true);
    (void)once;
#ifdef OODGAN_WITH_STRIPX8
    return zero_page() != nullptr && num_cus() > 0 && stripx8_init();
#else
    return zero_page() != nullptr && num_cus() > 0;
#endif
}

// strips are cut into segments only when there are fewer strips than CUs (one workgroup per CU: 152 KB of LDS); tiles of 4 rows
void segments(int B, int H, int W, int& tiles_x, int& tiles_y, int& seg_tiles, int& nseg) {
    tiles_y = (H + 3) / 4;
    tiles_x = (W + 31) / 32;
    const int ncu = num_cus() > 0 ? num_cus() : 256;
    const long strips = (long)B * tiles_x;
    int ns = (int)((ncu + strips - 1) / strips);
    if (ns < 1) ns = 1;
    seg_tiles = (tiles_y + ns - 1) / ns;
    if (seg_tiles < 8) seg_tiles = tiles_y < 8 ? tiles_y : 8;
    nseg = (tiles_y + seg_tiles - 1) / seg_tiles;
}

}  // namespace

// number of partial sums per (sample, channel) the x_fform == 2 kernel writes to fuse->part_r / part_t (and per (sample,
// 16-channel block) to part_max)
extern "C" int oodgan_conv3x3_xf_nparts(int B, int H, int W) {
    int tx, ty, st, ns;
    segments(B, H, W, tx, ty, st, ns);
    return tx * ns;
}

extern "C" int oodgan_conv3x3_xf_supported(int B, int K, int M, int H, int W) {
    if (!oodgan::bound_device_ok("conv3x3_xf_supported") || !stripx_init()) return 0;
    return (B > 0 && K == 32 && M == 32 && H >= 8 && W >= 32 && H % 4 == 0 && W % 32 == 0 && (long)H * W * 64 * 2 < (1L << 32)) ? 1 : 0;
}

#ifdef OODGAN_CLOCK_STAMP
// stamp build only: buf = n x 4 waves x 6 counters of workgroup blockIdx.x (last launch wins)
extern "C" int oodgan_debug_set_stripx_stamp_buffer(void* buf, long n) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stripx_stamp), &p, sizeof(p)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_stripx_stamp_n), &n, sizeof(n)) != hipSuccess) {
        oodgan::set_error("debug_set_stripx_stamp_buffer: hipMemcpyToSymbol failed");
        return OODGAN_E_LAUNCH;
    }
    return 0;
}
#endif

namespace oodgan {

int launch_s1_stripx(const oodgan_conv_args& a_in, const void* wpk16, const float* unscale, hipStream_t st) {
    StripX p;
    p.a = a_in;
    oodgan_conv_args& a = p.a;
    OODGAN_REQUIRE(oodgan_conv3x3_xf_supported(a.B, a.K, a.M, a.Hin, a.Win), "conv3x3 F-form input: needs K == M == 32, H %% 4 == 0, W %% 32 == 0");
    OODGAN_REQUIRE(a.mode == OODGAN_CONV_S1 && !a.x_sform && a.in_shift == nullptr && a.ys == nullptr && a.y != nullptr && a.groups <= 1,
                   "conv3x3 F-form input: mode S1, no shift, no S-form output");
    OODGAN_REQUIRE((reinterpret_cast<uintptr_t>(a.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0, "conv3x3 F-form input: 16-byte aligned x and y");
    if (a.out_pitch == 0) a.out_pitch = a.Win;
    p.out_plane = (long)a.Hin * a.out_pitch;
    p.w_unscale = unscale;
    p.zeros = zero_page();
    segments(a.B, a.Hin, a.Win, p.tiles_x, p.tiles_y, p.seg_tiles, p.nseg);
    p.Mp = (a.M + 63) / 64 * 64;
    p.nparts = p.tiles_x * p.nseg;
    const bool bwd = a.x_fform == 2;
    if (bwd) {
        OODGAN_REQUIRE(a.fuse != nullptr && a.dotx != nullptr && a.dotx_fform && a.dot_part != nullptr && a.in_scale == nullptr,
                       "conv3x3 x_fform 2: needs fuse (the activation backward), an F-form dotx, dot_part and no in_scale");
        OODGAN_REQUIRE(a.dot_nparts == p.nparts, "conv3x3 x_fform 2: dot_nparts %d != %d (oodgan_conv3x3_xf_nparts)", a.dot_nparts, p.nparts);
        OODGAN_REQUIRE(a.noise == nullptr && a.bias == nullptr && a.act == OODGAN_ACT_NONE && a.rgb_y == nullptr && !a.y_fform,
                       "conv3x3 x_fform 2: the input-gradient instance has no noise / bias / activation / ToRGB / F-form output");
        p.f = *a.fuse;
        OODGAN_REQUIRE(p.f.part_r && p.f.part_t && p.f.part_max && p.f.dscale && p.f.g_rgb && p.f.w_rgb && p.f.s_rgb,
                       "conv3x3 x_fform 2: incomplete fuse arguments (the ToRGB branch is mandatory: it is the only gradient source)");
        OODGAN_REQUIRE(p.f.noise == nullptr || p.f.noise_batch == 1 || p.f.noise_batch == a.B, "conv3x3 x_fform 2: noise_batch");
        OODGAN_REQUIRE((long)a.M * p.out_plane * 4 < (1L << 32), "conv3x3 x_fform 2: plane too large");
        a.fuse = nullptr;
    } else {
        OODGAN_REQUIRE(a.x_fform == 1 && a.dotx == nullptr && a.fuse == nullptr && a.y_fform && a.out_pitch == a.Win,
                       "conv3x3 x_fform 1: the forward instance writes a dense F-form y and takes no dotx / fuse");
        OODGAN_REQUIRE(a.act == OODGAN_ACT_NONE || a.act == OODGAN_ACT_LRELU, "conv3x3 x_fform 1: act must be none or lrelu");
        OODGAN_REQUIRE(a.rgb_y == nullptr || (a.rgb_w && a.rgb_s), "conv3x3 x_fform 1: the fused ToRGB needs rgb_w and rgb_s");
        p.f = oodgan_actbwd_fuse{};
    }
    const long nblk = (long)a.B * p.tiles_x * p.nseg;
    OODGAN_REQUIRE(nblk < (1L << 31), "conv3x3 F-form input: grid too large");
    OODGAN_REQUIRE(stripx_init(), "conv3x3 F-form input: initialisation failed (zero page / device query)");
    const uint4* w16 = reinterpret_cast<const uint4*>(wpk16);
    const dim3 grid((unsigned)nblk), block(256);
    if (bwd) {
        const bool seg = p.nseg > 1;
        if (a.x_hi_only) {
            count_dispatch(OODGAN_DC_STRIPX_G2);
            if (a.dot_actgrad) {
                if (seg) hipLaunchKernelGGL((conv_f16s_stripx_kernel<true, false, true, false, true, true>), grid, block, SXL<true>::SMEM, st, p, w16);
                else hipLaunchKernelGGL((conv_f16s_stripx_kernel<true, false, true, false, false, true>), grid, block, SXL<true>::SMEM, st, p, w16);
            } else {
                if (seg) hipLaunchKernelGGL((conv_f16s_stripx_kernel<true, false, false, false, true, true>), grid, block, SXL<true>::SMEM, st, p, w16);
                else hipLaunchKernelGGL((conv_f16s_stripx_kernel<true, false, false, false, false, true>), grid, block, SXL<true>::SMEM, st, p, w16);
            }
        } else if (a.dot_actgrad) {
            if (seg) hipLaunchKernelGGL((conv_f16s_stripx_kernel<true, false, true, false, true>), grid, block, SXL<true>::SMEM, st, p, w16);
            else hipLaunchKernelGGL((conv_f16s_stripx_kernel<true, false, true, false, false>), grid, block, SXL<true>::SMEM, st, p, w16);
        } else {
            if (seg) hipLaunchKernelGGL((conv_f16s_stripx_kernel<true, false, false, false, true>), grid, block, SXL<true>::SMEM, st, p, w16);
            else hipLaunchKernelGGL((conv_f16s_stripx_kernel<true, false, false, false, false>), grid, block, SXL<true>::SMEM, st, p, w16);
        }
    } else if (tunable(OODGAN_TUN_STRIPX_WAVES) == 8) {
#ifdef OODGAN_WITH_STRIPX8
        return launch_s1_stripx8_fwd(p, wpk16, st);      // two waves per SIMD, K split + specialised roles (experimental/conv_f16s_stripx8.hip)
#else
        set_error("conv3x3 F-form input: stripx_waves = 8 needs a library built with `make STRIPX8=1` (csrc/experimental/conv_f16s_stripx8.hip: measured "
                  "equal to the 4-wave kernel in round 5 and retired from the default build in round 6)");
        return OODGAN_E_ARG;
#endif
    } else if (a.rgb_y) hipLaunchKernelGGL((conv_f16s_stripx_kernel<false, true, false, true>), grid, block, SXL<false>::SMEM, st, p, w16);
    else hipLaunchKernelGGL((conv_f16s_stripx_kernel<false, false, false, true>), grid, block, SXL<false>::SMEM, st, p, w16);
    return check_launch("conv3x3_f16s_stripx");
}

}  // namespace oodgan
