/*
 * oodgan.h — C ABI of liboodgan_hip.so: the MI355X (gfx950) implementation of the hot path of
 * AbnerVictor/OOD-GAN-inversion (StyleGAN2 generator forward, SAMM/SAIM feature decomposition,
 * and the W+ latent-optimisation loop).  SURVEY.md §8(b) is the contract this header follows.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to fp32 (or int32 where stated) owned by the caller; the
 *     library never allocates, frees or retains caller memory (engine objects own only their
 *     private scratch, sized at creation);
 *   - tensors are dense NCHW unless a pitch is stated;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued asynchronously, no sync;
 *   - return value: OODGAN_OK (0) or a negative OODGAN_E_* code; no C++ exception crosses the
 *     boundary; oodgan_last_error() returns a thread-local message for the last failure;
 *   - a NULL optional pointer means "term absent" (the reference encodes that as an empty tensor,
 *     fused_bias_act_kernel.cu:62-63);
 *   - ONE HIP device per process (batch sharding = one process per GPU): per-kernel setup is done once, for the device that is
 *     current at the first conv call; a conv entry point (oodgan_conv3x3, oodgan_conv3x3_f16s, oodgan_modconv_f16,
 *     oodgan_conv3x3_xf_supported) called later with another device current returns OODGAN_E_ARG.
 *
 * Each entry names the reference interface it replaces (paths relative to /root/reference).
 */
#ifndef OODGAN_H
#define OODGAN_H

#ifdef __cplusplus
extern "C" {
#endif

#define OODGAN_OK 0
#define OODGAN_E_ARG (-1)      /* bad shape / null pointer / unsupported parameter */
#define OODGAN_E_LAUNCH (-2)   /* hipGetLastError() after a launch */
#define OODGAN_E_STATE (-3)    /* engine used in the wrong order */
#define OODGAN_E_NODEV (-4)    /* no HIP device */

int oodgan_version(void);
const char* oodgan_last_error(void);
/* number of HIP devices visible (0 on a CPU-only box); never throws */
int oodgan_device_count(void);
/* Dispatch tunables.  Each has a default, overridden ONCE (at first use) by an environment variable, and can be changed at
 * run time here — the hot path never calls getenv().  Names / environment / default:
 *   "s1_big_min_items" OODGAN_S1_BIG_MIN_ITEMS 128   work items (16x32-pixel tiles x images x 64-channel blocks) from which the
 *   "s2_big_min_items" OODGAN_S2_BIG_MIN_ITEMS 128   8-wave stride-1 / stride-2 / transposed kernels take a conv (unit tests lower
 *   "t2_big_min_items" OODGAN_T2_BIG_MIN_ITEMS 128   them to reach those kernels with small tensors, or raise them for the A/B)
 *   "blurt_strip"      OODGAN_BLURT_STRIP      1     0: the tile kernel instead of the strip walk in oodgan_act_bwd_blurT_sform_phases
 *   "blur_strip"       OODGAN_BLUR_STRIP       1     0: the tile kernel instead of the strip walk in oodgan_blur_act_fform / _sform_sep
 *   "upvb_waves"       OODGAN_UPVB_WAVES       12    form of oodgan_upconv_vblur_fform: 12 = one persistent 12-wave workgroup per CU, 6 / 4 = tile
 *                                                    kernels with two / three workgroups per CU (csrc/conv_f16s_upvb.hip)
 *   "fewout_quad"      OODGAN_FEWOUT_QUAD      1     0: oodgan_conv3x3_fewout2 keeps its one-pixel-per-thread form (the A/B of csrc/samm.hip's third form)
 *   "tiny_mid_max"     OODGAN_TINY_MID_MAX     1024  most positions (B*H*W) of a 16x16 / 32x32 stride-1 map the skinny-GEMM kernel takes when the caller
 *                                                    offers a workspace (the encoder trunk at batch 1-4; 8192 measured: +2.7 ms per batch of 8)
 *   "stripx_waves"     OODGAN_STRIPX_WAVES     4     the FORWARD F-form strip conv of the 1024² level (oodgan_conv_args.x_fform = 1): 4 = one wave per SIMD
 *                                                    (csrc/conv_f16s_stripx.hip, round 3); 8 = two waves per SIMD, the K loop split over a wave pair with
 *                                                    de-phased producer / finisher roles (csrc/experimental/conv_f16s_stripx8.hip, round 5: measured equal — retired from the default
 *                                                    build in round 6: only in a library built with `make STRIPX8=1`, OODGAN_E_ARG otherwise)
 * oodgan_set_tunable returns OODGAN_E_ARG for an unknown name; oodgan_get_tunable returns -1 for one. */
int oodgan_set_tunable(const char* name, long value);
long oodgan_get_tunable(const char* name);
/* Dispatch counters of oodgan_conv3x3_f16s: how many calls since load (or oodgan_dispatch_reset) went to the kernel family
 * `name` — "stripx" (conv_f16s_stripx.hip: F-form input, 1024² level of the W+ loop), "strip", "s1big", "s1v2", "s1pp", "tiny",
 * "t2big", "t2v2", "t2gen", "s2big", "s2v2", "s2gen", "upvb" (oodgan_upconv_vblur_fform), and the sub-counters of the fused epilogues:
 * "s1big_ys" (8-wave stride-1 launches that wrote `ys` / ToRGB partial sums), "s1big_g2" / "s2big_g2" / "stripx_g2" (input-gradient launches
 * that ran with x_hi_only, two matrix instructions per product), "s2big_xh" / "s1big_xh" (... on 32-byte hi-only input records, x_hi_only = 2), "s2big_fuse" (8-wave stride-2 launches with the fused activation
 * backward), "s2big_dotx_sform" (... that decoded `dotx` from a saved S-form).  Host-side, one relaxed atomic increment per call; the reference has no
 * counterpart (cuDNN picks its algorithm silently) — the parity tests use them to assert which kernel they pinned.
 * Returns -1 for an unknown name. */
long oodgan_dispatch_count(const char* name);
int oodgan_dispatch_reset(void);
/* Zero `bytes` bytes of device memory on `stream` (a fill kernel — graph-capture safe, unlike a memset node here): the accumulators / atomic-max slots the kernels expect zeroed.  The
 * host mirror uses it instead of torch.zeros inside the W+ loop, so that no torch kernel runs on the hot path. */
int oodgan_zero(void* p, long bytes, void* stream);

/* ------------------------------------------------------------------ launch plans (round 6) -- */
/* A plan is the list of kernel launches — kernel, grid, block, dynamic LDS, stream, by-value arguments (the packed descriptors the entry
 * points build from their argument structs) — that the calling thread made between oodgan_plan_record_begin and oodgan_plan_record_end
 * through ANY entry point of this library.  Recording does not change what the calls do (every launch is also issued).  oodgan_plan_run
 * re-issues the recorded launches `times` times, in order, on the streams they were recorded with: eager launches from C++ (not a
 * hipGraph), with none of the host work that built them — the ~170 launches of one W+ step cost one call instead of ~170 ctypes calls and
 * the Python around them (8-9 ms of host time per step against 10-11 ms of GPU time).
 * The caller guarantees what a hipGraph replay would need: every device buffer the recorded calls used is still allocated at the same
 * address and plays the same role (the host mirror records under a private allocator pool, oodgan/engine.py), and nothing a recorded call
 * derived on the host from per-step state changes (the W+ loop keeps its step counter and loss row on the device:
 * oodgan_adam_step_dev, oodgan_mse_fwd_bwd_row).  One plan is recorded and run by one thread at a time; different threads may record
 * different plans concurrently.  The reference has no counterpart (PyTorch's eager dispatcher; torch.cuda.graphs would be the analogue).
 * oodgan_plan_set_null_launch(1): process-wide, recorded AND direct launches become no-ops — host-cost probes only. */
void* oodgan_plan_create(void);
int oodgan_plan_destroy(void* plan);
int oodgan_plan_record_begin(void* plan);
long oodgan_plan_record_end(void* plan);       /* number of launches recorded, -1 on error */
long oodgan_plan_size(const void* plan);
int oodgan_plan_run(void* plan, int times);
int oodgan_plan_set_null_launch(int on);

/* ------------------------------------------------------------------ L1 custom ops ---------- */

/* y = scale * leaky_relu(x + noise_w[0]*noise[b,0,p] + bias[c], slope)
 * replaces: fused_bias_act(input,bias,refer,act=3,grad=0,alpha,scale) (src/ops/op/fused_bias_act.cpp:11-17)
 * and NoiseInjection.forward + FusedLeakyReLU (src/ops/StyleGAN/model.py:283-292,343-350).
 * x,y (B,C,HW); bias (C) or NULL; noise (noise_batch,1,HW) or NULL with noise_batch in {1,B};
 * noise_w device scalar (NULL = 1). */
int oodgan_bias_act_fwd(const float* x, const float* bias, const float* noise, const float* noise_w,
                        float* y, int B, int C, long HW, int noise_batch, float slope, float scale,
                        void* stream);

/* gx = gy * (y > 0 ? scale : slope*scale) — fused_bias_act(..., act=3, grad=1) with refer = out
 * (src/ops/op/fused_bias_act_kernel.cu:36-45, fused_act.py:25-58).  gbias (C) optional:
 * gbias[c] = sum_{b,p} gx (fused_act.py:36-41). */
int oodgan_bias_act_bwd(const float* gy, const float* y, float* gx, float* gbias, int B, int C, long HW,
                        float slope, float scale, void* stream);

/* upfirdn2d(input.reshape(-1,in_h,in_w,1), kernel, up_x,up_y,down_x,down_y,pad_x0,pad_x1,pad_y0,pad_y1)
 * replaces: upfirdn2d_op.upfirdn2d (src/ops/op/upfirdn2d.cpp:12-19) == upfirdn2d_native
 * (src/ops/op/upfirdn2d.py:160-193).  x (planes,in_h,in_w) row pitch in_pitch (elements; 0 = in_w);
 * y (planes,out_h,out_w) row pitch out_pitch; kernel (kh,kw) un-flipped (the op flips it).
 * Backward = same call with up/down swapped, flipped kernel and g_pad (upfirdn2d.py:115-120). */
int oodgan_upfirdn2d(const float* x, const float* kernel, float* y, int planes, int in_h, int in_w,
                     int in_pitch, int out_pitch, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                     int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* Blur(pad) of the up-sampling StyledConv fused with its tail: y = act(upfirdn2d(x,k,pad=(pad0,pad1)) +
 * noise_w*noise + bias)   (src/ops/StyleGAN/model.py:255-258,343-350).  x (B,C,in_h,in_w) row pitch in_pitch. */
int oodgan_blur_bias_act(const float* x, const float* kernel, float* y, int B, int C, int in_h, int in_w,
                         int in_pitch, int kh, int kw, int pad0, int pad1, const float* bias, const float* noise,
                         int noise_batch, const float* noise_w, int act, void* stream);

/* ------------------------------------------------------------------ A1 style affine -------- */

/* Batched EqualLinear(512->Ci, bias_init=1) for any number of modulation layers at once:
 * s[b,r] = scale * sum_k wcat[r,k]*latent[b,row_lat[r],k] + bcat[r]*lr_mul
 * replaces: ModulatedConv2d.modulation(style) (src/ops/StyleGAN/model.py:148-158,223,236), 26 calls
 * per generator forward.  latent (B,L,S); wcat (R,S); bcat (R) or NULL; row_lat int32 (R) or NULL (all 0). */
int oodgan_style_affine_fwd(const float* latent, const float* wcat, const float* bcat, const int* row_lat,
                            float* s, int B, int L, int S, int R, float scale, float lr_mul, void* stream);
/* The MFMA path (R%16==0 and S%16==0) requires every aligned group of 16 rows to share one latent
 * index (row_lat[r] == row_lat[r & ~15]); the generator's channel counts guarantee it.
 * glat[b,l,k] = scale * sum_{r in [lat_start[l], lat_start[l+1])} gs[b,r]*wcat[r,k]  (overwrites glat);
 * rows must be grouped by latent index; lat_start int32 (L+1). */
int oodgan_style_affine_bwd(const float* gs, const float* wcat, const int* lat_start, float* glat,
                            int B, int L, int S, int R, float scale, void* stream);
/* EqualLinear with fused_lrelu (mapping network layer, model.py:148-151): y = sqrt2*lrelu(scale*x@W^T + b*lr_mul) */
int oodgan_equal_linear(const float* x, const float* w, const float* b, float* y, int B, int in_dim, int out_dim,
                        float scale, float lr_mul, int activate, void* stream);
/* G independent EqualLinear layers side by side — x (B,G,I), w (G,O,I), bias (G,O) or NULL -> y (B,G,O); bit-identical to G calls of
 * oodgan_equal_linear (the final linears of the GradualStyleBlock heads, psp_encoders.py:31-34, as ONE launch) */
int oodgan_equal_linear_grouped(const float* x, const float* w, const float* b, float* y, int B, int G, int in_dim, int out_dim,
                                float scale, float lr_mul, int activate, void* stream);
/* PixelNorm (model.py:11-16): y = x * rsqrt(mean_k x^2 + 1e-8) */
int oodgan_pixel_norm(const float* x, float* y, int B, int S, void* stream);

/* ------------------------------------------------------------------ A2/A3 modulated conv --- */

/* wsq[co,ci] = sum_k w[co,ci,k]^2   (weight preparation for the demodulation coefficients) */
int oodgan_weight_sqsum(const float* w, float* wsq, int Co, int Ci, int KK, void* stream);
/* d[b,co] = rsqrt(scale^2 * sum_ci s[b,ci]^2 * wsq[co,ci] + 1e-8)  — model.py:239-241 factorised
 * (SURVEY.md Appendix A).  s row stride s_stride, d row stride d_stride (elements). */
int oodgan_demod_fwd(const float* s, int s_stride, const float* wsq, float* d, int d_stride,
                     int B, int Ci, int Co, float scale, void* stream);
/* gs[b,ci] += s[b,ci]*(-scale^2) * sum_co r[b,co]*d[b,co]^2*wsq[co,ci],  r = sum_p gy*y (y demodulated) */
int oodgan_demod_bwd(const float* s, int s_stride, const float* wsq, const float* d, int d_stride,
                     const float* r, float* gs, int gs_stride, int B, int Ci, int Co, float scale, void* stream);

/* Pack a (Co,Ci,3,3) conv weight (optionally scaled) into the K-major layout the MFMA kernels
 * stream: wpk[k][tap][Mp], Mp = round_up(M,64), zero padded.
 *   transpose=0: k=ci, m=co (forward);  transpose=1: k=co, m=ci (input-gradient);
 *   flip=1: tap -> 8-tap (adjoint of a stride-1 correlation). */
int oodgan_pack_conv3x3(const float* w, float* wpk, int Co, int Ci, float scale, int transpose, int flip,
                        void* stream);

#define OODGAN_CONV_S1 0   /* 3x3, stride 1, zero pad 1:            (B,K,H,W)       -> (B,M,H,W)       */
#define OODGAN_CONV_T2 1   /* 3x3 transposed, stride 2, pad 0:       (B,K,H,W)       -> (B,M,2H+1,2W+1) */
#define OODGAN_CONV_S2 2   /* 3x3, stride 2, no pad:                 (B,K,2H+1,2W+1) -> (B,M,H,W)       */

#define OODGAN_ACT_NONE 0
#define OODGAN_ACT_LRELU 1   /* sqrt2 * leaky_relu(.,0.2) */
#define OODGAN_ACT_PRELU 2   /* per-channel slope */

struct oodgan_actbwd_fuse;
typedef struct oodgan_conv_args {
    const float* x;          /* (B,K,Hin,Win), row pitch in_pitch */
    const float* wpk;        /* packed by oodgan_pack_conv3x3 */
    const float* in_scale;   /* (B,K) stride in_scale_stride, or NULL: staged value = x*in_scale+in_shift */
    const float* in_shift;   /* (B,K) same stride, or NULL (applied to in-bounds samples only) */
    const float* out_scale;  /* (B,M) stride out_scale_stride, or NULL */
    const float* bias;       /* (M) or NULL */
    const float* noise;      /* (noise_batch,1,Hout,Wout) or NULL */
    const float* noise_w;    /* device scalar or NULL (=1) */
    const float* slope;      /* (M) PReLU slopes for OODGAN_ACT_PRELU */
    const float* dotx;       /* (B,M,Hout,Wout) or NULL: dot_part = sum_p acc*dotx BEFORE out_scale */
    float* dot_part;         /* (B,M,dot_nparts) partial sums; reduce with oodgan_reduce_parts */
    float* y;                /* (B,M,Hout,Wout), row pitch out_pitch */
    int B, K, M, Hin, Win;   /* Hin/Win: logical input size */
    int in_pitch, out_pitch; /* elements; 0 = dense */
    int in_scale_stride, out_scale_stride;
    int noise_batch;
    int mode, act;
    int dot_nparts;          /* out: must equal oodgan_conv3x3_nparts(...) */
    const float* in_mul2;    /* device {unscale, scale} pair or NULL: the staged input is multiplied by scale and the
                                accumulators by unscale (power-of-two range control of the split-f16 kernels,
                                produced by oodgan_absmax_scale; ignored by the exact-fp32 kernel) */
    int x_sform;             /* 1: x is an S-form buffer (oodgan_to_sform / a producer's `ys`): already scaled and split,
                                in_scale/in_shift must be NULL (split-f16 kernels only) */
    void* ys;                /* optional S-form output of (activated y) * ys_scale[b,m] for the next conv, or NULL */
    const float* ys_scale;   /* (B,M) stride ys_scale_stride, or NULL */
    int ys_scale_stride;
    /* optional fused ToRGB partial: rgb_y[b,k,p] = rgb_scale * sum_m rgb_w[k,m] * rgb_s[b,m] * act(y)[b,m,p]  (ToRGB.forward without
     * bias / skip, model.py:363-372; finish with oodgan_rgb_finish).  rgb_y NULL = off.  The split-f16 strip kernel (mode S1,
     * 16 < K,M <= 32) writes the complete sums; the 8-wave stride-1 kernel (round 4, together with `ys`) writes one partial sum per
     * 64-channel block: rgb_y is then (ceil(M/64), B, 3, Hout, Wout) and oodgan_rgb_finish_parts adds the blocks in order. */
    const float* rgb_w;      /* (3,M) */
    const float* rgb_s;      /* (B,*) stride rgb_s_stride */
    float* rgb_y;            /* (B,3,Hout,Wout) dense */
    int rgb_s_stride;
    float rgb_scale;
    const struct oodgan_actbwd_fuse* fuse;   /* optional fused activation backward (mode S2, split-f16, S-form input), or NULL */
    int dot_actgrad;         /* 1 (mode S1 with dotx, where oodgan_conv3x3_s1_actgrad_supported): dotx is the OUTPUT of the
                                up-sampling StyledConv below, y is the gradient w.r.t. it, and the epilogue applies that
                                layer's FusedLeakyReLU backward (fused_act.py:25-58): y <- y * (dotx>0 ? sqrt2 : 0.2*sqrt2).
                                oodgan_act_bwd_blurT_sform_phases then takes this y with out == NULL. */
    int groups;              /* G > 1: grouped convolution (nn.Conv2d(groups=G) semantics, e.g. the 18 GradualStyleBlock heads of the
                                e4e encoder run side by side, src/ops/e4e/encoders/psp_encoders.py:14-34): x is (B, G*K, Hin, Win), the
                                packed weights hold G*Mg output channels (M = G*Mg, Mg %% 64 == 0) of K inputs each, and output
                                channel m convolves input channels [g*K, (g+1)*K) with g = m / Mg; in_scale / in_shift are then
                                (B, G*K).  0 / 1 = dense.  Supported by oodgan_conv3x3_f16s for fp32 NCHW input in mode S2. */
    int y_fform;             /* 1: y is written in F-form — [B][ceil(M/16)][Hout][Wout][16] fp32, one 64-byte record per pixel and
                                16-channel block — instead of NCHW.  Only the split-f16 strip kernel (mode S1, 16 < K,M <= 32, no
                                dotx): the LAST styled conv of the generator in the W+ loop, whose output is read back by nothing
                                but its own activation backward (oodgan_act_bwd_sform_f), a pure 16-byte-per-lane stream then. */
    int x_fform;             /* != 0: x is fp32 in F-form ([B][2][Hin][Win][16]) and the kernel converts it to its split-f16 operand
                                itself (csrc/conv_f16s_stripx.hip; mode S1, K == M == 32, Hin %% 4 == 0, Win %% 32 == 0, see
                                oodgan_conv3x3_xf_supported) — the 1024² level of the W+ loop, where a separate S-form copy of the
                                tensor costs a write and a read of 1 GB each per step:
                                1 (forward): staged value = x * in_scale[b,k] * in_mul2[1]; y is written in F-form (y_fform = 1);
                                  bias / noise / lrelu / fused ToRGB as in the strip kernel.
                                2 (input gradient): x is the saved OUTPUT `out` of the StyledConv whose activation backward produces
                                  this conv's input; `fuse` describes that backward exactly as for oodgan_act_bwd_sform
                                  (g = s_rgb*t, g_pre = g*act'(out), staged value = g_pre*dscale*mul2[1]; fuse->ys unused) and
                                  receives part_r / part_t (B,32,nparts) and part_max (B*2*nparts floats) with
                                  nparts = oodgan_conv3x3_xf_nparts(B,Hin,Win) — also the dot_nparts of this instance; dotx (F-form,
                                  dotx_fform = 1) is mandatory, y NCHW. */
    int dotx_fform;          /* 1: dotx is in F-form (x_fform == 2 only) */
    int dotx_sform;          /* 1 (mode S2 with `fuse` only, round 4): dotx is not an fp32 tensor but the S-form a conv's epilogue wrote for its consumer
                                (`ys` of the 8-wave stride-1 kernel, y == NULL): the values are x * dotx_scale[b,m] as f16 pairs; the fused epilogue
                                decodes them (hi + lo) / dotx_scale.  The saved activation of the 128² ... 512² conv layers inside the W+ loop then
                                exists once, not twice. */
    const float* dotx_scale; /* (B,M) stride dotx_scale_stride: the scale that went into that S-form (style x range scale of its consumer) */
    int dotx_scale_stride;
    void* workspace;         /* optional scratch of workspace_bytes >= oodgan_conv3x3_tiny_workspace(...) bytes (the K-split partial tiles:
                                written by the first launch, read by the finishing one — uninitialised is fine) owned by ONE stream: with it the 4x4 / 8x8 layers (mode S1 with S-form input, mode S2
                                with phase-split S-form input, K >= 64) run as a skinny GEMM over the batch with a K split
                                (csrc/conv_f16s_tiny.hip); NULL: the tile kernels */
    long workspace_bytes;
    unsigned* ys_vmax;       /* optional, with `ys` from the 8-wave stride-1 kernel (oodgan_conv3x3_s1_ys_supported): (B x OODGAN_VMAX_SLOTS) float bit patterns,
                                max |activated y * ys_scale| of every sample atomically maxed into a slot — the forward range control of the conv
                                that reads `ys` (as the `vmax` argument of the S-form producers), or NULL */
    int x_hi_only;           /* 1 (round 6, precision 'f16s-g2'; split-f16 kernels with an S-form / F-form input AND `dotx`, i.e. the input-gradient
                                instances — ignored elsewhere): the lo half of the INPUT operand is dropped, x_hi * (w_hi + w_lo): two matrix
                                instructions per product instead of three.  The back-propagated gradient is then rounded to f16 (2^-11 relative,
                                zero mean, independent per element) before each contraction while the weights keep their 22 bits; the reference
                                has no counterpart (torch autograd runs the backward in fp32, model.py:233-274 through conv2d's backward) —
                                tests bound dL/dW+ against the float64 reference and the 100-step loss curve against the reference Adam loop.
                                2 (with dotx; mode S2 with a phase-split S-form input of a shape oodgan_conv3x3_s2_fuse_supported accepts, or mode S1 with an
                                S-form input on the 8-wave kernel, oodgan_conv3x3_s1_actgrad_supported-like shapes): as 1, and x holds 32-byte
                                hi-only records — what oodgan_act_bwd_blurT_sform_phases_hi / oodgan_actbwd_fuse.ys_hi_only write (half the bytes
                                written and read). */
} oodgan_conv_args;

/* Fused epilogue of the stride-2 input-gradient conv (csrc/conv_f16s_s2big.hip).  The conv's result IS the gradient
 * g_feat w.r.t. the output `out` (= args->dotx) of the StyledConv below, so the kernel continues with that layer's
 * oodgan_act_bwd_sform arithmetic (autograd of NoiseInjection + FusedLeakyReLU merged with the ToRGB branch,
 * src/ops/StyleGAN/model.py:283-292,343-372) on its accumulators: neither g_feat nor g_pre goes to HBM.
 *   t = rgb_scale * sum_k w_rgb[k,m]*g_rgb[b,k,p];  g = g_feat + s_rgb[b,m]*t;  g_pre = g * (out>0 ? sqrt2 : 0.2*sqrt2)
 *   ys       <- S-form of g_pre * dscale[b,m] * mul2[1]                         (input of that layer's input-gradient conv)
 *   part_r   <- partial sum_p g_pre*y_cv,  part_t <- partial sum_p out*t       (B,M,nparts), nparts = args->dot_nparts
 *   part_max <- partial max |g_pre*dscale| (nmax floats, zero-initialised by the caller; feeds oodgan_absmax_scale_check)
 * args->y may be NULL (g_feat is not stored); args->dotx / dot_part are mandatory; M %% 32 == 0. */
typedef struct oodgan_actbwd_fuse {
    const float* g_rgb;      /* (B,3,H,W) or NULL */
    const float* w_rgb;      /* (3,M) */
    const float* s_rgb;      /* (B,*) stride s_rgb_stride */
    const float* noise;      /* (noise_batch,H,W) or NULL */
    const float* noise_w;
    const float* bias;       /* (M) or NULL */
    const float* dscale;     /* (B,*) stride dscale_stride */
    const float* mul2;       /* device {unscale, scale} carried from the previous optimisation step */
    void* ys;
    float* part_r;
    float* part_t;           /* NULL when g_rgb is NULL */
    float* part_max;
    int s_rgb_stride, noise_batch, dscale_stride;
    float rgb_scale;
    long nmax;               /* entries of part_max: B * tiles * ceil(M/64) * 8 */
    int ys_hi_only;          /* 1 (round 6, precision 'f16s-g2'; oodgan_conv3x3_f16s mode S2 `fuse` only): ys receives 32-byte hi-only records — record
                                r of a (b, 16-channel block) plane at byte r*32 of that plane, the plane stride unchanged — for a consumer that never
                                reads the lo halves: mode S1 with x_hi_only = 2 on the 8-wave kernel.  The buffer must be one that only ever holds
                                hi-only records (its zero border lives at the hi-only addresses). */
} oodgan_actbwd_fuse;

/* Implicit-GEMM 3x3 convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 * replaces: the grouped F.conv2d / F.conv_transpose2d of ModulatedConv2d.forward
 * (src/ops/StyleGAN/model.py:247-272) — with input-side modulation (in_scale = style) and
 * output-side demodulation (out_scale) instead of B materialised weight copies — and the dense
 * Conv2d 3x3 of bottleneck_IR (src/ops/e4e/encoders/helpers.py:439-444). */
int oodgan_conv3x3(const oodgan_conv_args* args, void* stream);

/* Split-f16 variant (default of the engine): every operand is split v = hi + lo in f16 and each product
 * costs three v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulate): fp32-equivalent accuracy
 * (~1e-6 relative) at up to 16/3 the fp32 matrix rate.  Same arguments and epilogues as
 * oodgan_conv3x3; args->wpk must come from oodgan_pack_conv3x3_f16s, which also writes unscale2[2] =
 * {2^-e, 2^e}: the weights are stored times 2^e (max |w| in [512,1024)) and the kernel multiplies by 2^-e. */
long oodgan_pack_conv3x3_f16s_bytes(int Co, int Ci, int transpose);
int oodgan_pack_conv3x3_f16s(const float* w, void* wpk16, float* unscale2, int Co, int Ci, float scale, int transpose,
                             int flip, void* stream);
int oodgan_conv3x3_f16s(const oodgan_conv_args* args, const float* unscale2, void* stream);
int oodgan_conv3x3_f16s_nparts(int mode, int Hin, int Win);   /* dot_nparts expected by oodgan_conv3x3_f16s */
int oodgan_conv3x3_f16s_nparts2(int mode, int Hin, int Win, int x_sform);   /* same, for an S-form input */
/* 1 when mode S2 with an S-form input of this shape accepts oodgan_conv_args.fuse (the 8-wave kernel of csrc/conv_f16s_s2big.hip) */
int oodgan_conv3x3_s2_fuse_supported(int B, int K, int M, int Hin, int Win);
/* 1 when mode S2 with a PHASE-SPLIT S-form input of G*K channels accepts oodgan_conv_args.groups = G (K inputs per group, M = G*Mg outputs,
 * Mg % 128 == 0): the 8-wave kernel, whose channel block picks its group's K channels; plain epilogue (bias / activation) only.
 * Outputs of 8x8 and below (down to 1x1) with a workspace (oodgan_conv3x3_tiny_workspace) take groups too: the skinny-GEMM kernel. */
int oodgan_conv3x3_s2_grouped_supported(int B, int K, int M, int groups, int Hin, int Win);
/* 1 when mode S1 with an S-form input and dotx of this shape accepts oodgan_conv_args.dot_actgrad (strip / 8-wave kernels)
 * and oodgan_act_bwd_blurT_sform_phases(out = NULL) exists for the (H/2, W/2) layer below */
int oodgan_conv3x3_s1_actgrad_supported(int B, int K, int M, int H, int W);
/* 1 when mode S1 with an S-form input of this shape writes `ys` from the 8-wave kernel's registers; `y` may then be NULL (only the S-form
 * of the activated output x ys_scale is produced: conv -> activation -> conv chains without the fp32 tensor in between) */
int oodgan_conv3x3_s1_ys_supported(int B, int K, int M, int H, int W);
/* oodgan_conv_args.x_fform: 1 when the shape is supported, and the number of partial sums per (sample, channel) that the
 * x_fform == 2 instance writes to fuse->part_r / part_t (per (sample, 16-channel block) to fuse->part_max) */
int oodgan_conv3x3_xf_supported(int B, int K, int M, int H, int W);
int oodgan_conv3x3_xf_nparts(int B, int H, int W);
/* bytes of oodgan_conv_args.workspace the skinny-GEMM kernel of the 4x4 / 8x8 layers needs for this call (0: not applicable) */
long oodgan_conv3x3_tiny_workspace(int mode, int B, int K, int M, int Hin, int Win);

/* S-form activations (csrc/sform.hpp): per pixel and 16-channel block one 64-byte record {hi[16], lo[16]} f16 of the
 * value already multiplied by the consumer's scale, with a zero border and tile padding, so that the split-f16 convs
 * fetch their halo'd tiles as contiguous runs by LDS-DMA.  Buffers must be zero-initialised once (border).
 * oodgan_to_sform converts an fp32 NCHW tensor: value = (x*scale[b,c] + shift[b,c])*mul2[1].
 * Every forward producer of an S-form (oodgan_to_sform, oodgan_blur_act_sform, oodgan_torgb_fwd_sform) takes an optional
 * `vmax` (B x OODGAN_VMAX_SLOTS unsigned, float bit patterns, atomically maxed into one of the sample's slots): the
 * largest |value| it wrote for each sample — the input of the forward range control below. */
#define OODGAN_VMAX_SLOTS 64
long oodgan_sform_bytes(int B, int C, int H, int W);
/* value = (x*scale[b,c] + shift[b,c]) * mul2[1]; scale / shift / mul2 may be NULL.  The shift only reaches image pixels — the
 * border of the S-form stays zero, as the zero padding of a conv applied AFTER an affine normalisation requires (the SAMM
 * bottlenecks: InstanceNorm folded into scale and shift, src/ops/SAMM/helpers.py bottleneck_IR). */
int oodgan_to_sform(const float* x, const float* scale, int scale_stride, const float* shift, int shift_stride,
                    const float* mul2, void* out, int B, int C, int H, int W, int in_pitch, unsigned* vmax, void* stream);
/* Phase-split S-form for the stride-2 conv (mode S2 with x_sform): the (2H+1)x(2W+1) input is stored as its four
 * parity images G[py][px][i][j] = x[2i+py][2j+px], each in S-form without border, so that the stride-2 conv
 * becomes stride-1 taps on contiguous runs.  H,W = OUTPUT size of the S2 conv. */
long oodgan_sform_phases_bytes(int B, int C, int H, int W);
/* The way back: fp32 NCHW y[b,c,p] = (hi + lo) / scale[b,c] of an S-form buffer (scale NULL = 1) — for an activation saved only as the
 * S-form of its consumer (oodgan_conv_args.dotx_sform) when a step needs the plain tensor after all */
int oodgan_from_sform(const void* xs, const float* scale, int scale_stride, float* y, int B, int C, int H, int W, void* stream);
/* Fused producer: g (B,C,2H,2W) -> upfirdn2d(g, kernel, pad=(2,2)) (the adjoint of Blur(pad=(1,1)), src/ops/op/upfirdn2d.py:115-120)
 * * scale[b,c] * mul2[1], written phase-split.  kernel (4,4): the op correlates with the FLIPPED kernel like upfirdn2d. */
int oodgan_blurT_to_sform_phases(const float* g, const float* kernel, const float* scale, int scale_stride,
                                 const float* mul2, void* out, int B, int C, int H, int W, void* stream);
int oodgan_to_sform_phases(const float* x, const float* scale, int scale_stride, const float* mul2, void* out, int B,
                           int C, int H, int W, int in_pitch, void* stream);
/* the same for the FORWARD use of the stride-2 conv (nn.Conv2d(K, M, 3, stride 2, padding 1)): x is the unpadded (B,C,2H,2W) tensor, pitch in_pitch
 * (0: 2W); the zero row / column on the top / left of the (2H+1) x (2W+1) image the conv reads are produced here, not by a padded copy */
int oodgan_to_sform_phases_padtl(const float* x, const float* scale, int scale_stride, const float* mul2, void* out, int B,
                                 int C, int H, int W, int in_pitch, void* stream);
int oodgan_conv3x3_nparts(int mode, int Hin, int Win);
/* ---- forward range control of the split-f16 path (csrc/fwd_range.hip).  ModulatedConv2d.forward in fp32
 * (src/ops/StyleGAN/model.py:233-274) has no range limit; an S-form record (hi+lo f16) holds |v| < 65504.  Every styled
 * conv l therefore carries one power-of-two scale per sample q[l][b] with max|x*s|*q in [512,1024): the producers get the
 * style block s_sc = s*q, the conv epilogue the demodulation block d_sc = d/q (both exact).
 *   oodgan_absmax_scaled:    max_k vmax[b][k] = max_{c,p} |x[b,c,p]*s[b,c]| (atomic max into a zeroed array; non-finite -> inf)
 *   oodgan_fwd_range_update: n entries (layer-major [l][b], OODGAN_VMAX_SLOTS slots each); flag != NULL (carry mode): vmax was measured on values scaled
 *                            by q -> flag |= 1 if it left [1, 2^15) (round 5; 2^-8 before: lo halves near the maximum were f16 subnormals there), |= 2 if non-finite; next q from vmax/q.
 *                            flag == NULL (exact mode): vmax is the true max, q is set from it.  vmax is zeroed.
 *   oodgan_fwd_range_plan:   s_sc[b,r] = s_all[b,r]*q[row_layer[r]][b] for rows [row0,row0+nrows) (row_layer < 0: copy),
 *                            d_sc[b,r] = d_all[b,r]/q[drow_layer[r]][b] for rows [drow0,drow0+ndrows). */
int oodgan_absmax_scaled(const float* x, const float* s, int s_stride, unsigned* vmax, int B, int C, long HW, void* stream);
int oodgan_fwd_range_update(unsigned* vmax, float* q, int* flag, int n, void* stream);
int oodgan_fwd_range_plan(const float* s_all, const float* d_all, const int* row_layer, const int* drow_layer, const float* q,
                          float* s_sc, float* d_sc, int B, int R, int DR, int row0, int nrows, int drow0, int ndrows,
                          void* stream);

/* ---- fused forward producer (csrc/fwd_producers.hip): the tail of the up-sampling StyledConv in one pass —
 * Blur(pad=(1,1)) of the transposed-conv output z (B,C,2H+1,in_pitch) (model.py:72-88,199-205), + noise_w*noise + bias,
 * leaky-ReLU*sqrt2 (model.py:283-292,343-350) -> y (B,C,2H,2W) fp32 AND, if ys != NULL, the S-form of y*ys_scale[b,c]
 * (the next conv's input).  in_pitch %% 4 == 0 (what oodgan_conv3x3 mode T2 produces). */
int oodgan_blur_act_sform(const float* z, const float* kernel, float* y, void* ys, const float* ys_scale, int ys_scale_stride,
                          const float* bias, const float* noise, int noise_batch, const float* noise_w, int act, int B, int C,
                          int H, int W, int in_pitch, unsigned* vmax, void* stream);
/* The same tail writing y in F-form ([B][C/16][2H][2W][16] fp32, C %% 16 == 0) and NO S-form: for the layer whose following conv
 * converts its input itself (oodgan_conv_args.x_fform = 1).  ys_scale (the next conv's style x range scale) only enters vmax. */
int oodgan_blur_act_fform(const float* z, const float* kernel, float* y, const float* ys_scale, int ys_scale_stride,
                          const float* bias, const float* noise, int noise_batch, const float* noise_w, int act,
                          int B, int C, int H, int W, int in_pitch, unsigned* vmax, int kernel_rank_one, void* stream);
/* kernel_rank_one: 1 when the caller knows the 4x4 kernel to be an outer product (Blur's [1,3,3,1] x [1,3,3,1] is): selects the
 * strip-walk kernel (one horizontal and one vertical 4-tap pass); 0: the tile kernel, which tests the taps itself. */
/* ---- the whole up-sampling StyledConv in ONE pass (csrc/conv_f16s_upvb.hip, round 4): conv_transpose2d(stride 2, pad 0) of the S-form
 * input xs (K channels, H x W: x * style, split) -> Blur(pad = (1,1)) -> * out_scale[b,m] (demodulation) + noise_w*noise + bias ->
 * leaky-ReLU*sqrt2 -> y in F-form ([B][M/16][2H][2W][16] fp32).  replaces: ModulatedConv2d.forward's upsample branch + Blur +
 * NoiseInjection + FusedLeakyReLU of one StyledConv (src/ops/StyleGAN/model.py:199-205,247-258,283-292,343-350) — i.e.
 * oodgan_conv3x3_f16s(mode T2) followed by oodgan_blur_act_fform, without the (2H+1) x (2W+1) intermediate in memory.
 * The blur kernel must be rank one, kf[a][b] = kv[a]*kh[b] (kf = the FLIPPED 4x4 kernel, as upfirdn2d applies it).  Its vertical pass
 * is folded into the weights by the caller — two 3x3 weight sets, one per output-row parity py:
 *     Wv[py][d+1][kx] = sum over (a, ky) with py + a - 1 - ky == 2d of kv[a] * W[ky][kx],  d = -1, 0, 1
 * each packed with oodgan_pack_conv3x3_f16s (transpose = flip = 0) into ONE buffer, the second set wset_bytes behind the first;
 * unscale4 = the two {2^-e, 2^e} pairs of the packs; kh4 = the four horizontal taps kh[0..3] (device).  The horizontal pass runs on
 * the accumulators.  ys_scale / vmax as in oodgan_blur_act_fform (forward range control of the following conv).
 * Shapes: K %% 16 == 0, M %% 32 == 0, H >= 4, W >= 30 (oodgan_upconv_vblur_supported).  Twice the matrix work of the transposed conv:
 * meant for the highest level (64 -> 32 channels, 512² -> 1024²), where the two passes it replaces are bound by memory. */
int oodgan_upconv_vblur_supported(int B, int K, int M, int H, int W);
int oodgan_upconv_vblur_fform(const void* xs, const void* wpk2, long wset_bytes, const float* unscale4, const float* kh4,
                              const float* out_scale, int out_scale_stride, const float* bias, const float* noise, int noise_batch,
                              const float* noise_w, int act, const float* ys_scale, int ys_scale_stride, unsigned* vmax, float* y,
                              int B, int K, int M, int H, int W, void* stream);
/* oodgan_blur_act_sform with the same promise about the kernel (C %% 16 == 0 and 2W >= 64 for the strip walk; the tile kernel otherwise) */
int oodgan_blur_act_sform_sep(const float* z, const float* kernel, float* y, void* ys, const float* ys_scale, int ys_scale_stride,
                              const float* bias, const float* noise, int noise_batch, const float* noise_w, int act, int B, int C,
                              int H, int W, int in_pitch, unsigned* vmax, int kernel_rank_one, void* stream);

/* ---- fused backward producers (csrc/bwd_producers.hip): oodgan_act_bwd_fused's arithmetic (autograd of NoiseInjection +
 * FusedLeakyReLU merged with the ToRGB branch, src/ops/StyleGAN/model.py:283-292,343-372) written directly as the
 * S-form input of the next matrix kernel — value g_pre * dscale[b,c] * mul2[1] — so the fp32 g_pre never goes to HBM.
 * mul2 = {unscale, scale} is the range scale measured on the PREVIOUS optimisation step; part_max receives this pass's
 * per-block max|g_pre| and oodgan_absmax_scale_check (a) sets flag bit0 if max*scale left [2^-8, 2^15), bit1 on a
 * non-finite value, (b) overwrites state = {2^-e, 2^e} with max*2^e in [512,1024) for the next step.
 * part_r / part_t: (B,C,nparts) partial sums to be reduced with oodgan_reduce_parts. */
int oodgan_act_bwd_sform_nparts(int H, int W);
int oodgan_act_bwd_sform(const float* g_feat, const float* out, const float* noise, int noise_batch, const float* noise_w,
                         const float* bias, const float* g_rgb, const float* w_rgb, const float* s_rgb, int s_rgb_stride,
                         float rgb_scale, const float* dscale, int dscale_stride, const float* mul2, void* ys, float* part_r,
                         float* part_t, float* part_max, int B, int C, int H, int W, void* stream);
/* oodgan_act_bwd_sform with `out` in F-form (oodgan_conv_args.y_fform) and no g_feat: the activation backward of the LAST styled
 * conv (its only gradient is the ToRGB branch).  Same arithmetic per element, same nparts; loads and stores are 16 bytes per lane
 * on contiguous runs, no LDS transpose.  oodgan_from_fform converts an F-form tensor back to NCHW. */
int oodgan_act_bwd_sform_f(const float* out_f, const float* noise, int noise_batch, const float* noise_w, const float* bias,
                           const float* g_rgb, const float* w_rgb, const float* s_rgb, int s_rgb_stride, float rgb_scale,
                           const float* dscale, int dscale_stride, const float* mul2, void* ys, float* part_r, float* part_t,
                           float* part_max, int B, int C, int H, int W, void* stream);
int oodgan_from_fform(const float* f, float* y, int B, int C, int H, int W, void* stream);
/* same, followed by blur^T (adjoint of Blur(pad=(1,1)), src/ops/op/upfirdn2d.py:115-120) and the phase split of
 * oodgan_blurT_to_sform_phases; H,W = size of the up-conv's INPUT, the tensors are (B,C,2H,2W). */
int oodgan_act_bwd_blurT_nparts(int H, int W);
/* out == NULL: g_feat already is g_pre = dx * act'(out), made by the conv above (oodgan_conv_args.dot_actgrad, dotx = out).
 * The demodulation-gradient sum of the layer, r = sum_p g_pre*y_cv with y_cv = act^-1(out) - noise_w*noise - bias, then
 * splits as  sum_p dx*out - sum_p g_pre*(noise_w*noise + bias)  (g_pre * act^-1(out) = dx * out on either branch):
 * part_r receives the partials of the SECOND term (negative sign included) and the first is out_scale[b,c] * dot of the
 * conv above — combine with an oodgan_reduce_job that has part2 / scale2.  Exists where
 * oodgan_act_bwd_blurT_pre_supported(H, W) says so. */
int oodgan_act_bwd_blurT_pre_supported(int H, int W);
int oodgan_act_bwd_blurT_sform_phases(const float* g_feat, const float* out, const float* noise, int noise_batch,
                                      const float* noise_w, const float* bias, const float* g_rgb, const float* w_rgb,
                                      const float* s_rgb, int s_rgb_stride, float rgb_scale, const float* dscale,
                                      int dscale_stride, const float* mul2, const float* kernel, void* out_phases,
                                      float* part_r, float* part_t, float* part_max, int B, int C, int H, int W, void* stream);
/* The same without a ToRGB branch and with HI-ONLY output (round 6, precision 'f16s-g2'): 32-byte records holding the hi f16 halves of the 16
 * channels — record r of a phase plane at byte r*32 of the plane's first half of the SAME buffer geometry (oodgan_sform_phases_bytes) — for a
 * consumer that never reads the lo halves: oodgan_conv3x3_f16s with oodgan_conv_args.x_hi_only = 2.  Halves the bytes this producer writes and
 * that conv reads.  Exists where oodgan_act_bwd_blurT_hi_supported(H, W) says so (the strip walk). */
int oodgan_act_bwd_blurT_hi_supported(int H, int W);
/* 1 when oodgan_conv3x3_f16s (mode S1, S-form input, dotx) of this shape takes x_hi_only = 2 (the 8-wave stride-1 kernel) */
int oodgan_conv3x3_s1_xh_supported(int B, int K, int M, int H, int W);
int oodgan_act_bwd_blurT_sform_phases_hi(const float* g_feat, const float* out, const float* noise, int noise_batch, const float* noise_w,
                                         const float* bias, const float* dscale, int dscale_stride, const float* mul2, const float* kernel,
                                         void* out_phases, float* part_r, float* part_max, int B, int C, int H, int W, void* stream);
int oodgan_absmax_scale_check(const float* part, long n, float* state, int* flag, void* stream);

/* ---- batched tail of the W+ backward (csrc/bwd_tail.hip): per-layer jobs that only feed the style-gradient accumulator,
 * one launch per kind; `jobs` is a HOST array (copied into the kernel arguments).  Same arithmetic and per-output
 * summation order as oodgan_reduce_parts[_cols] / oodgan_demod_bwd / oodgan_absmax_scale_check. */
typedef struct oodgan_reduce_job {
    const float* part;       /* (B,C,nparts) */
    float* out;              /* out[b*out_stride + c] (+)= sum_j part[b,c,j]  [+ scale2[b*scale2_stride + c] * sum_j part2[b,c,j]] */
    int B, C, nparts, out_stride, accumulate;
    const float* part2;      /* (B,C,nparts2) or NULL */
    const float* scale2;     /* (B,*) stride scale2_stride */
    int nparts2, scale2_stride;
} oodgan_reduce_job;
typedef struct oodgan_demod_bwd_job {
    const float* s; const float* wsq; const float* d; const float* r; float* gs;
    int s_stride, d_stride, gs_stride, B, Ci, Co;
    float scale;
} oodgan_demod_bwd_job;
typedef struct oodgan_demod_fwd_job {
    const float* s; const float* wsq; float* d;
    int s_stride, d_stride, B, Ci, Co;
    float scale;
} oodgan_demod_fwd_job;
typedef struct oodgan_scale_check_job {
    const float* part; long n; float* state;
} oodgan_scale_check_job;
int oodgan_reduce_batch(const oodgan_reduce_job* jobs, int njobs, void* stream);
int oodgan_demod_bwd_batch(const oodgan_demod_bwd_job* jobs, int njobs, void* stream);
int oodgan_demod_fwd_batch(const oodgan_demod_fwd_job* jobs, int njobs, void* stream);   /* oodgan_demod_fwd for all layers */
int oodgan_absmax_scale_check_batch(const oodgan_scale_check_job* jobs, int njobs, int* flag, void* stream);

/* ---- fp16 modulated conv for the high-resolution, low-channel layers (BASELINE.json configs[4] / SURVEY §8 C5) ----
 * ModulatedConv2d.forward, plain 3x3 (src/ops/StyleGAN/model.py:233-245,268-274) + NoiseInjection + FusedLeakyReLU
 * (model.py:283-292,343-350) in f16 operands / fp32 accumulate / f16 result.  Activations live in "H-form":
 *   X[b][ceil(C/16)][Hp][Wp][16 x f16]  (32-byte record per pixel and 16-channel block; pixel (y,x) at [y+1][x+1];
 *   zero border + tile padding as the S-form; buffers must be zero-initialised once).
 * As in the reference the modulation and demodulation are folded into per-sample weights (model.py:236-241), which
 * oodgan_modconv_f16_pack builds from the fp32 master weight (M,K,3,3) and the style (B,K).  K, M <= 32.
 * `act` given to the pack and to the conv must agree: the leaky ReLU's sqrt(2) gain is folded into the weights. */
long oodgan_hform_bytes(int B, int C, int H, int W);
int oodgan_to_hform(const float* x, void* out, int B, int C, int H, int W, void* stream);      /* fp32 NCHW -> H-form */
int oodgan_from_hform(const void* in, float* y, int B, int C, int H, int W, void* stream);    /* H-form -> fp32 NCHW */
long oodgan_modconv_f16_wbytes(int B, int M, int K);
int oodgan_modconv_f16_pack(const float* weight, const float* style, int style_stride, float scale, int demodulate,
                            int act, void* wpk, int B, int M, int K, void* stream);
/* the same with the modulation EqualLinear inside (ModulatedConv2d's `self.modulation`, model.py:219-223,236):
 * style[b,k] = sum_j latent[b,j] * mod_weight[k,j] / sqrt(S) + mod_bias[k] — one launch for affine + modulate + demodulate + pack */
int oodgan_modconv_f16_pack_affine(const float* weight, const float* latent, int latent_stride, const float* mod_weight,
                                   const float* mod_bias, int S, float scale, int demodulate, int act, void* wpk, int B, int M,
                                   int K, void* stream);
/* y = act(conv3x3(x, w[b]) + noise_w*noise + bias); noise (noise_batch,H,W) fp32 or NULL, act = OODGAN_ACT_NONE|LRELU */
int oodgan_modconv_f16(const void* x, const void* wpk, const float* noise, int noise_batch, const float* noise_w,
                       const float* bias, int act, void* y, int B, int K, int M, int H, int W, void* stream);
/* out[i] (+)= sum_j part[i,j]  (deterministic two-stage reductions) */
int oodgan_reduce_parts(const float* part, float* out, long rows, int nparts, int accumulate, void* stream);
/* the same for a (B,C,nparts) array, written to (accumulate=0) or added to (1) out[b*out_stride + c]: the style gradient
 * of a layer lands directly in its column block of the (B, sum Ci) accumulator (autograd's += of model.py:236-241's s) */
int oodgan_reduce_parts_cols(const float* part, float* out, int B, int C, int nparts, int out_stride, int accumulate,
                             void* stream);

/* ------------------------------------------------------------------ A5 ToRGB -------------- */

/* y[b,c,p] = sum_ci scale*w[c,ci]*s[b,ci]*x[b,ci,p] + bias[c] + upfirdn2d(skip, k4*4, up=2, pad=(2,1))[b,c,p]
 * replaces: ToRGB.forward (src/ops/StyleGAN/model.py:363-372) incl. Upsample (:30-48).
 * x (B,Ci,H,W); w (3,Ci); s (B,Ci) stride s_stride; skip (B,3,H/2,W/2) or NULL; kernel (4,4) already x4. */
int oodgan_torgb_fwd(const float* x, const float* w, const float* s, int s_stride, const float* bias,
                     const float* skip, const float* kernel, float* y, int B, int Ci, int H, int W,
                     float scale, void* stream);
/* The same ToRGB, additionally writing the S-form (split-f16 layout of oodgan_to_sform) of x*ys_scale[b,ci] for the
 * up-sampling ModulatedConv2d that reads the same feature map next (model.py:557-576: to_rgb and the next conv1 share
 * `out`).  ys: S-form buffer of (B,Ci,H,W); ys_scale (B,*) stride ys_scale_stride or NULL.  Ci % 16 == 0, W % 4 == 0. */
int oodgan_torgb_fwd_sform(const float* x, const float* w, const float* s, int s_stride, const float* bias,
                           const float* skip, const float* kernel, float* y, void* ys, const float* ys_scale,
                           int ys_scale_stride, int B, int Ci, int H, int W, float scale, unsigned* vmax, void* stream);
/* y[b,k,p] = partial[b,k,p] + bias[k] + upfirdn2d(skip, k4*4, up=2, pad=(2,1))[b,k,p]: second half of ToRGB.forward when the
 * three colour sums came out of the conv kernel's epilogue (oodgan_conv_args.rgb_y).  y may alias partial. */
int oodgan_rgb_finish(const float* partial, const float* bias, const float* skip, const float* kernel, float* y, int B, int H,
                      int W, void* stream);
/* the same for `nparts` partial sums (nparts, B, 3, H, W), added in the order of their index (oodgan_conv_args.rgb_y of the 8-wave kernel) */
int oodgan_rgb_finish_parts(const float* partial, int nparts, const float* bias, const float* skip, const float* kernel, float* y, int B,
                            int H, int W, void* stream);

/* feature_modulation(gen_feats, conditions, None, mod_type) (src/ops/StyleGAN/model.py:588-610; called from
 * Generator.forward :558-566 and StyleGAN2Generator.forward, stylegan2_arch.py:583-588, for cond_type != 'NOISE'), clss = 1:
 * mode 0 'SFT'  y = x*(1 + c0) + c1;  1 'ADD'  y = x + c1 (c0 may be NULL);  2 'FUSE'  y = x + c1*sigmoid(c0).  n elements. */
int oodgan_feature_modulation(const float* x, const float* c0, const float* c1, float* y, long n, int mode, void* stream);

/* Backward through (bias + noise + lrelu*sqrt2) of one StyledConv, merged with the ToRGB branch that
 * reads the same feature (build-defined W+ loop):
 *   t[b,c,p]   = rgb_scale * sum_k w_rgb[k,c]*g_rgb[b,k,p]                (0 if g_rgb NULL)
 *   g          = g_feat[b,c,p] (0 if NULL) + s_rgb[b,c]*t
 *   g_pre      = g * (out>0 ? sqrt2 : 0.2*sqrt2)
 *   y_cv       = (out>0 ? out/sqrt2 : out/(0.2*sqrt2)) - noise_w*noise[b,p] - bias[c]
 *   part_r[b,c,j]   = partial sum_p g_pre*y_cv    (demodulation gradient)
 *   part_rgb[b,c,j] = partial sum_p out*t         (ToRGB style gradient)
 * nparts = oodgan_act_bwd_nparts(HW). */
int oodgan_act_bwd_fused(const float* g_feat, const float* out, const float* noise, int noise_batch,
                         const float* noise_w, const float* bias, const float* g_rgb, const float* w_rgb,
                         const float* s_rgb, int s_rgb_stride, float rgb_scale, float* g_pre,
                         float* part_r, float* part_rgb, int B, int C, long HW, void* stream);
int oodgan_act_bwd_nparts(long HW);
/* same, additionally part_max[b,c,j] = partial max |g_pre| * |dscale[b,c]| (dscale = the layer's demodulation block, stride
 * dscale_stride, NULL = 1): the largest value the consumer's S-form will hold before its range scale (feeds oodgan_absmax_scale) */
int oodgan_act_bwd_fused_max(const float* g_feat, const float* out, const float* noise, int noise_batch,
                             const float* noise_w, const float* bias, const float* g_rgb, const float* w_rgb,
                             const float* s_rgb, int s_rgb_stride, float rgb_scale, float* g_pre,
                             float* part_r, float* part_rgb, float* part_max, const float* dscale, int dscale_stride, int B,
                             int C, long HW, void* stream);
/* out2 = {2^-e, 2^e} with e chosen so that max_i |part[i]| * 2^e lies in [512,1024) (e = 0 for an all-zero or
 * non-finite input): the power-of-two scale that keeps a tensor inside the f16 range of the split-f16 kernels. */
int oodgan_absmax_scale(const float* part, long n, float* out2, void* stream);
/* the same, and `part` is zeroed afterwards: a persistent slot array needs no fill launch between measurements */
int oodgan_absmax_scale_clear(float* part, long n, float* out2, void* stream);

/* ------------------------------------------------------------------ A9 loss / optimiser ---- */

/* loss[b] = mean_{c,p} (img-target)^2 (per image, deterministic), gimg = grad_mul*2*(img-target)/(C*HW).
 * grad_mul is the (power-of-two) loss scale that keeps the back-propagated values O(1) for the split-f16
 * kernels; the caller divides the final latent gradient by it (exact).
 * anchors: basicsr MSELoss (BasicSR/basicsr/losses/losses.py:58-83). part: (B,nparts) scratch. */
int oodgan_mse_fwd_bwd(const float* img, const float* target, float* gimg, float* part, float* loss,
                       int B, long CHW, float grad_mul, void* stream);
int oodgan_mse_nparts(long CHW);
/* the same with the losses written to row min(row_dev[0], nrows-1) of loss_table (nrows, B): row_dev is the W+ loop's device step
 * counter (the value oodgan_adam_step_dev increments at the END of a step, i.e. the zero-based index of the running step), so a
 * recorded step (oodgan_plan_run, hipGraph replay) fills the loss table row by row without a host-side pointer per step */
int oodgan_mse_fwd_bwd_row(const float* img, const float* target, float* gimg, float* part, float* loss_table,
                           const int* row_dev, int nrows, int B, long CHW, float grad_mul, void* stream);
/* torch.optim.Adam step (no weight decay, no amsgrad), step index t>=1 given by the host:
 * anchors: get_optimizer (src/models/OOD_faceGAN_model.py:398-400). */
int oodgan_adam_step(float* w, const float* g, float* m, float* v, long n, float lr, float beta1,
                     float beta2, float eps, int t, void* stream);
/* same with the step index on the device: increments t_dev[0] first, then uses it (hipGraph-replayable W+ step) */
int oodgan_adam_step_dev(float* w, const float* g, float* m, float* v, long n, float lr, float beta1,
                         float beta2, float eps, int* t_dev, void* stream);

/* ------------------------------------------------------------------ X1 LPIPS(alex) term ---- */
/* The perceptual term of the inversion loss (north_star: "W+ Adam steps against LPIPS/L2"; reference call site
 * src/losses/lpips_loss.py:13-34: lpips.LPIPS(net='alex')(pred, target, normalize=True) on images mapped from min_max to [0,1]).
 * PARITY UNPINNED: the `lpips` package and its weights are not in the reference tree; these ops restate the published
 * algorithm and are checked against oracle/lpips_cpu.py on seeded weights.  Host mirror: oodgan/lpips.py.
 *
 * Stride-1 2-D convolution, exact fp32 on the matrix cores, NCHW:  y = conv(x, w, pad) [+ bias] [ReLU] [+ add] [* (mask > 0)]
 * (torch.nn.Conv2d of AlexNet's feature stack, and — with flipped / transposed weights and pad' = ks-1-pad — its input gradient, where
 * `add` is the gradient arriving at the same tensor from its LPIPS tap and `mask` the ReLU output below).  x (B,K,Hin,Win);
 * wpk [K][ks*ks][round_up(M,64)] (zero padded; tap = ky*ks+kx, correlation as torch); y/add/mask (B,M,Hin+2pad-ks+1,Win+2pad-ks+1);
 * ks in {3,5}.  AlexNet's first conv (11x11, stride 4, pad 2) runs as ks=3, pad=0 on the 48-channel space-to-depth image of
 * oodgan_lpips_prep. */
int oodgan_conv2d_s1(const float* x, const float* wpk, const float* bias, const float* add, const float* mask, float* y, int B, int K,
                     int M, int Hin, int Win, int ks, int pad, int relu, void* stream);
/* y <- (y + add) * (mask > 0) in place, n elements (16-byte aligned): the epilogue of oodgan_conv2d_s1's backward use as a pass of its own, behind an
 * input-gradient conv that ran on oodgan_conv3x3_f16s (the 3x3 layers of the stack on the split-f16 matrix kernels, oodgan/lpips.py) */
int oodgan_add_mask(float* y, const float* add, const float* mask, long n, void* stream);
/* nn.MaxPool2d(kernel_size=3, stride=2) on (planes,H,W) -> (planes,(H-3)/2+1,(W-3)/2+1), and its backward merged with what surrounds
 * it in LPIPS: gx = (scatter of gy to each window's first maximum + add) * (x > 0), x being a ReLU output (add NULL = 0).  idx (optional, one byte
 * per OUTPUT element): the forward records each window's argmax (dy*3+dx, first maximum in row-major order as torch keeps it) and the backward reads it
 * instead of re-scanning the windows. */
int oodgan_maxpool3s2_fwd(const float* x, float* y, unsigned char* idx, long planes, int H, int W, void* stream);
int oodgan_maxpool3s2_bwd(const float* x, const float* gy, const float* add, const unsigned char* idx, float* gx, long planes, int H, int W, void* stream);
/* image (B,3,H,W) -> conv1 operand (B,48,H/4+1,W/4+1): v = a*x + b0 (min_max -> [-1,1]: lpips_loss.py:27-29 followed by lpips'
 * normalize=True), lpips.ScalingLayer (v - shift[c]) / scale[c], zero pad 2, 4x4 space-to-depth (channel c*16 + dy*4 + dx).
 * shift3 / scale3 are HOST arrays of 3 floats.  H, W multiples of 4. */
int oodgan_lpips_prep(const float* img, float* out48, int B, int H, int W, float a, float b0, const float* shift3, const float* scale3,
                      void* stream);
/* gimg (B,3,H,W) += coef * a / scale[c] * depth_to_space(g48): the gradient w.r.t. the image from the gradient w.r.t. the conv1 operand */
int oodgan_lpips_img_grad(const float* g48, float* gimg, int B, int H, int W, float a, float coef, const float* scale3, void* stream);
/* One LPIPS tap on features (B,C,HW).  mode 0: out = f0 / (sqrt(sum_c f0^2) + 1e-10)  (lpips.normalize_tensor; the target's, once).
 * mode 1: part[b][blk] = partial spatial sums of d = sum_c w[c] (n0_c - n1_c)^2 (lin layer on the squared difference), blk <
 * oodgan_lpips_head_nparts(HW), and out = coef * d(d)/d(f0) (unmasked; the consumer applies the ReLU mask).
 * mode 2: mode 1 with out * (f0 > 0): the deepest tap, whose gradient nothing else joins. */
int oodgan_lpips_head(const float* f0, const float* n1, const float* w, float* out, float* part, int B, int C, long HW, float coef,
                      int mode, void* stream);
int oodgan_lpips_head_nparts(long HW);
/* lpips[b] = sum_taps (sum_blk parts[t][b][blk]) / hw[t]  (spatial_average + sum over the taps) -> row min(row_dev[0], nrows-1) of
 * table (nrows,B) (row_dev NULL: row 0).  parts / nparts / hw: HOST arrays of ntaps <= 5 entries. */
int oodgan_lpips_finish(const float* const* parts, const int* nparts, const long* hw, int ntaps, float* table, const int* row_dev,
                        int nrows, int B, void* stream);

/* ------------------------------------------------------------------ A7/A10 SAMM / SAIM ----- */

/* per-(b,c) mean and rstd (biased variance, eps) of x (B,C,HW): nn.InstanceNorm2d statistics
 * (src/ops/SAMM/helpers.py:88, e4e/encoders/helpers.py:93-95). stats (B,C,2) = {mean, rstd}. */
int oodgan_instnorm_stats(const float* x, float* stats, int B, int C, long HW, float eps, void* stream);
/* scale/shift so that IN_affine(x) = x*sc + sh: sc = rstd*gamma, sh = beta - mean*rstd*gamma (gamma/beta NULL = 1/0) */
int oodgan_instnorm_coeffs(const float* stats, const float* gamma, const float* beta, float* sc, float* sh,
                           int B, int C, void* stream);
/* y = x*sc[b,c] + sh[b,c] (+ res) */
int oodgan_affine_apply(const float* x, const float* sc, const float* sh, const float* res, float* y,
                        int B, int C, long HW, void* stream);
/* oodgan_affine_apply followed by oodgan_instnorm_stats of its result (bit-identical statistics), in one pass: stats (B, C, 2) */
int oodgan_affine_apply_stats(const float* x, const float* sc, const float* sh, const float* res, float* y, float* stats, int B, int C,
                              long HW, float eps, void* stream);
/* AlignNet input (src/ops/SAMM/helpers.py:97-101): out[:, C:] = IN(enc) and
 *   diff != 0 (diff_fAndg=True, every shipped config):  out[:, :C] = IN(gen) - IN(enc)
 *   diff == 0 (diff_fAndg=False, round 5 / ABI 108):    out[:, :C] = IN(gen);        stats from oodgan_instnorm_stats */
int oodgan_align_input(const float* gen, const float* enc, const float* st_gen, const float* st_enc,
                       float* out, int diff, int B, int C, long HW, void* stream);
/* oodgan_align_input followed by oodgan_instnorm_stats of its (B, 2C, H, W) result (bit-identical statistics), in one pass: stats (B, 2C, 2) */
int oodgan_align_input_stats(const float* gen, const float* enc, const float* st_gen, const float* st_enc, float* out, float* stats, int diff,
                             int B, int C, long HW, float eps, void* stream);
/* dense 1x1 conv with optional bias: y[b,m,p] = sum_k w[m,k]*x[b,k,p] + bias[m]
 * replaces feats_conv (OOD_faceGAN_e4e_arch.py:70-75) and the AlignNet shortcut (helpers.py:431-434) */
int oodgan_conv1x1(const float* x, const float* w, const float* bias, float* y, int B, int K, int M, long HW,
                   void* stream);
/* squeeze-excitation gate of bottleneck_IR_SE (SEModule.forward, src/ops/e4e/encoders/helpers.py:60-76):
 * gate[b,c] = sigmoid(sum_j w2[c,j] * relu(sum_k w1[j,k] * mean[b,k])), mean = stats[b,k,0] of oodgan_instnorm_stats;
 * w1 (Cr,C), w2 (C,Cr): the 1x1 conv weights of fc1 / fc2; C <= 1024, Cr <= 64 */
int oodgan_se_gate(const float* stats, const float* w1, const float* w2, float* gate, int B, int C, int Cr, void* stream);
/* small direct 3x3 conv (K,M <= 8), pad 1, with optional in scale/shift (B,K) and PReLU */
int oodgan_conv3x3_small(const float* x, const float* w, const float* in_sc, const float* in_sh,
                         const float* slope, float* y, int B, int K, int M, int H, int W, void* stream);
/* 3x3 conv, stride 1, pad 1, from many input channels to M <= 4 outputs, exact fp32, K split over workgroups with a
 * deterministic combine: the head conv 2C -> 3 of AlignNet's second bottleneck (src/ops/SAMM/helpers.py:58-60,
 * bottleneck_IR(2C, 3) of src/ops/e4e/encoders/helpers.py:439-444) incl. the preceding InstanceNorm as in_sc / in_sh (B,K)
 * (shift on in-bounds samples only) and the following PReLU (slope (M) or NULL).  w is the raw (M,K,3,3) weight.
 * part: workspace (B, oodgan_conv3x3_fewout_ksplit(B,K,H,W), M, H, W) floats. */
int oodgan_conv3x3_fewout_ksplit(int B, int K, int H, int W);
int oodgan_conv3x3_fewout(const float* x, const float* w, const float* in_sc, const float* in_sh, const float* slope,
                          float* part, float* y, int B, int K, int M, int H, int W, void* stream);
/* The same conv from a (K, 9, 4) transposed copy of the weight (K % 8 == 0), optionally together with the 1x1 conv of the same bottleneck's
 * shortcut (w11t (K, 4): (M2 <= 4, K) transposed, applied to the RAW x; y2 (B, M2, H, W), part2 like part) in one pass over x. */
int oodgan_conv3x3_fewout2(const float* x, const float* wt, const float* w11t, const float* in_sc, const float* in_sh, const float* slope,
                           float* part, float* part2, float* y, float* y2, int B, int K, int M, int M2, int H, int W, void* stream);
/* AlignNet head (helpers.py:104-107): ch0,1 -> tanh*scale ; ch2 -> sigmoid */
int oodgan_align_head(const float* x, float* y, int B, long HW, float scale, void* stream);
/* SPM_Warp.add / upsample_add (helpers.py:129-147) with new_PRM (:62-77):
 * mode 0 (add):          dx,dy = clip(acc+cur, +-scale); a = clip(cur_a*acc_a + acc_a*(1-acc_a),0,1)
 * mode 1 (upsample_add): dx,dy = cur; pa = bicubic_ac(prev_a -> HxW); a = clip(cur_a*pa + pa*(1-pa),0,1)
 * acc/cur/out (B,3,H,W); prev (B,3,Hp,Wp). */
int oodgan_field_compose(const float* acc, const float* cur, const float* prev, float* out, int B, int H, int W,
                         int Hp, int Wp, float scale, int mode, void* stream);
/* grid_sample(target, identity+field[:, :2]) (bilinear, zeros, align_corners=False) then
 * lerp with field[:,2]: y = warped*a + target*(1-a)   (helpers.py:168-177) */
int oodgan_warp_blend(const float* target, const float* field, float* y, int B, int C, int H, int W, void* stream);
/* y = cond + noise_w*noise ; then bias + lrelu — the conditioned-layer epilogue
 * (OOD_faceGAN_e4e_arch.py:239-242 + model.py:292,348) is oodgan_bias_act_fwd on cond. */
/* blending_mask + blend (OOD_faceGAN_e4e_arch.py:315-347): alpha = compose(bilinear-upsampled alpha
 * channels of up to 4 fields, coarse->fine), clip; out = alpha*x + gen*(1-alpha).
 * fields: HOST array of nfields (<=4) device pointers, fields[i] (B,3,sizes[i],sizes[i]); sizes: HOST array;
 * alpha_out (B,1,S,S) ; out (B,3,S,S). gen/x/out may be NULL (mask only). */
int oodgan_mask_blend(const float* const* fields, const int* sizes, int nfields, const float* x, const float* gen,
                      float* alpha_out, float* out, int B, int S, void* stream);
/* F.interpolate(x, size, mode='nearest') (run_ood_faceGAN_inversion.py:80-83): src = floor(dst*in/out) */
int oodgan_resize_nearest(const float* x, float* y, int planes, int Hin, int Win, int Hout, int Wout,
                          int out_pitch, int out_xoff, void* stream);
/* F.interpolate(x, size, mode='bilinear', align_corners=False) */
/* F.interpolate(mode='bicubic', align_corners=True) to (Hout,Wout), optionally + add (same shape as the result):
 * `_upsample_add` of the e4e encoder's FPN (src/ops/e4e/encoders/helpers.py:504-521). */
int oodgan_resize_bicubic_ac(const float* x, const float* add, float* y, int planes, int Hin, int Win, int Hout, int Wout,
                             void* stream);
/* AdaptiveAvgPool2d((Hout, Wout)): `face_pool` of the ReStyle / FeatureStyle variants
 * (src/archs/OOD_faceGAN_restyle_arch.py:89, 292-303; 1024 -> 256) and the 3x3 pooled descriptors of `fs_encoder_v2`
 * (src/ops/FeatureStyle/feature_style_encoder.py:42,55-66). */
int oodgan_avgpool(const float* x, float* y, int planes, int Hin, int Win, int Hout, int Wout, void* stream);
int oodgan_resize_bilinear(const float* x, float* y, int planes, int Hin, int Win, int Hout, int Wout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OODGAN_H */
