"""Thin functional wrappers: torch ROCm tensors in (used purely as device buffers), HIP kernels
through the C ABI, torch tensors out.  Names/arguments mirror the reference's op layer
(src/ops/op/upfirdn2d.py:149, src/ops/op/fused_act.py:92) where one exists."""
import ctypes
import os
import math

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_PRELU, CONV_S1, CONV_S2, CONV_T2, ConvArgs, check

SQRT2 = 2 ** 0.5


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_DEV_INDEX = None


def _stream_handle():
    """Raw handle of the current HIP stream of this process's device (one device per process, include/oodgan.h).  The host mirror issues
    ~700 launches per model(x): at batch 1 the eager call is bound by the host, and `torch.cuda.current_stream().cuda_stream` — a Stream
    object per launch — was a measurable part of it."""
    global _DEV_INDEX
    if _RAW_STREAM is None:
        return torch.cuda.current_stream().cuda_stream
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return _RAW_STREAM(_DEV_INDEX)


def _stream():
    return ctypes.c_void_p(_stream_handle())


def zeros(*shape, device, dtype=torch.float32):
    """A zeroed device tensor without a torch kernel: ``torch.empty`` + ``oodgan_zero`` (a fill kernel of the library on the current stream).  Used
    for the accumulators of the W+ loop (torch tensors are containers only on the hot path)."""
    t = torch.empty(*shape, device=device, dtype=dtype)
    check(_lib.lib().oodgan_zero(_p(t), t.numel() * t.element_size(), _stream()), 'zero')
    return t


class Cols:
    """A column block [off, off+n) of a dense (B, R) matrix, passed by pointer + row stride (no copy)."""

    def __init__(self, base, off, n):
        self.base, self.off, self.n = base, off, n
        self.shape = (base.shape[0], base.shape[1])   # shape[1] is the ROW STRIDE seen by the kernels
        self.is_cuda, self.dtype = base.is_cuda, base.dtype

    def data_ptr(self):
        return self.base.data_ptr() + 4 * self.off

    def dense(self):
        return self.base[:, self.off:self.off + self.n]


def _dev(t, name='input'):
    if isinstance(t, Cols):
        return t
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f'{name} must be a ROCm (cuda) tensor: the HIP path has no CPU fallback')
    if t.dtype != torch.float32:
        raise RuntimeError(f'{name} must be float32, got {t.dtype}')
    return t.contiguous()


def _opt(t, name):
    return None if t is None else _dev(t, name)


# ----------------------------------------------------------------------------- L1 ops
def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0), device='hip', out_pitch=0):
    """reference signature: upfirdn2d(input, kernel, up=1, down=1, pad=(0,0), device='cpu')
    (src/ops/op/upfirdn2d.py:149-157).  ``device`` is accepted and ignored: this always runs the HIP
    kernel on the tensor's GPU."""
    x = _dev(input)
    k = _dev(kernel, 'kernel')
    B, C, H, W = x.shape
    kh, kw = k.shape
    p0, p1 = int(pad[0]), int(pad[1])
    oh = (H * up + p0 + p1 - kh) // down + 1
    ow = (W * up + p0 + p1 - kw) // down + 1
    if oh <= 0 or ow <= 0:
        raise RuntimeError(f'upfirdn2d: empty output {oh}x{ow}')
    y = torch.empty(B, C, oh, out_pitch if out_pitch else ow, device=x.device, dtype=torch.float32)
    if B * C == 0:
        return y
    check(_lib.lib().oodgan_upfirdn2d(_p(x), _p(k), _p(y), B * C, H, W, 0, out_pitch, kh, kw, up, up, down, down, p0, p1, p0,
                                      p1, _stream()), 'upfirdn2d')
    return y


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=SQRT2, device='hip'):
    """reference: fused_leaky_relu(input, bias, negative_slope=0.2, scale=2**0.5, device='cpu')
    (src/ops/op/fused_act.py:92-96).  Bias indexes dim 1."""
    x = _dev(input)
    shp = x.shape
    B, C = shp[0], shp[1] if x.ndim > 1 else 1
    HW = x.numel() // max(B * C, 1)
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    b = _opt(bias, 'bias')
    check(_lib.lib().oodgan_bias_act_fwd(_p(x), _p(b), None, None, _p(y), B, C, HW, 1, float(negative_slope), float(scale),
                                         _stream()), 'bias_act_fwd')
    return y


def fused_leaky_relu_backward(grad_output, out, negative_slope=0.2, scale=SQRT2, need_bias_grad=False):
    """FusedLeakyReLUFunctionBackward (src/ops/op/fused_act.py:25-58): returns (grad_input, grad_bias)."""
    g = _dev(grad_output, 'grad_output')
    o = _dev(out, 'out')
    B, C = g.shape[0], g.shape[1]
    HW = g.numel() // (B * C)
    gx = torch.empty_like(g)
    gb = torch.empty(C, device=g.device, dtype=torch.float32) if need_bias_grad else None
    check(_lib.lib().oodgan_bias_act_bwd(_p(g), _p(o), _p(gx), _p(gb), B, C, HW, float(negative_slope), float(scale),
                                         _stream()), 'bias_act_bwd')
    return gx, gb


def bias_noise_act(x, bias=None, noise=None, noise_weight=None, negative_slope=0.2, scale=SQRT2):
    """NoiseInjection + FusedLeakyReLU in one pass (src/ops/StyleGAN/model.py:283-292,343-350)."""
    x = _dev(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    nz = _opt(noise, 'noise')
    nb = 1 if nz is None else nz.shape[0]
    check(_lib.lib().oodgan_bias_act_fwd(_p(x), _p(_opt(bias, 'bias')), _p(nz), _p(_opt(noise_weight, 'noise_weight')), _p(y),
                                         B, C, HW, nb, float(negative_slope), float(scale), _stream()), 'bias_act_fwd')
    return y


def blur_bias_act(x, kernel, pad, bias=None, noise=None, noise_weight=None, act=True, in_hw=None, in_pitch=0):
    """Blur(pad) + noise + bias + lrelu*sqrt2 (src/ops/StyleGAN/model.py:255-258,343-350).  ``x`` may
    be a pitched buffer (B,C,in_h,in_pitch) with logical width in_hw[1]."""
    x = _dev(x)
    k = _dev(kernel, 'kernel')
    B, C = x.shape[0], x.shape[1]
    H, W = in_hw if in_hw is not None else (x.shape[2], x.shape[3])
    kh, kw = k.shape
    oh, ow = H + pad[0] + pad[1] - kh + 1, W + pad[0] + pad[1] - kw + 1
    y = torch.empty(B, C, oh, ow, device=x.device, dtype=torch.float32)
    nz = _opt(noise, 'noise')
    nb = 1 if nz is None else nz.shape[0]
    check(_lib.lib().oodgan_blur_bias_act(_p(x), _p(k), _p(y), B, C, H, W, in_pitch, kh, kw, int(pad[0]), int(pad[1]),
                                          _p(_opt(bias, 'bias')), _p(nz), nb, _p(_opt(noise_weight, 'nw')),
                                          ACT_LRELU if act else ACT_NONE, _stream()), 'blur_bias_act')
    return y


_MOD_TYPES = {'SFT': 0, 'ADD': 1, 'FUSE': 2}


def feature_modulation(gen_feats, conditions, clss=None, mod_type='SFT'):
    """reference: feature_modulation(gen_feats, conditions, clss=None, mod_type='SFT') (src/ops/StyleGAN/model.py:588-610);
    ``clss`` other than None is not used by any caller.  conditions = [c0, c1] tensors of gen_feats' shape."""
    if clss is not None:
        raise NotImplementedError('feature_modulation: clss is None at every call site of the reference')
    if mod_type not in _MOD_TYPES:
        raise NotImplementedError(f'unknown mod_type {mod_type}')
    x = _dev(gen_feats)
    c1 = _dev(conditions[1], 'conditions[1]')
    c0 = None if (mod_type == 'ADD' or conditions[0] is None) else _dev(conditions[0], 'conditions[0]')
    if c1.shape != x.shape or (c0 is not None and c0.shape != x.shape):
        c1 = c1.expand_as(x).contiguous()
        c0 = None if c0 is None else c0.expand_as(x).contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().oodgan_feature_modulation(_p(x), _p(c0), _p(c1), _p(y), x.numel(), _MOD_TYPES[mod_type], _stream()), 'feature_modulation')
    return y


# ----------------------------------------------------------------------------- style path
def style_affine(latent, wcat, bcat=None, row_lat=None, lr_mul=1.0):
    """s[b,r] = (1/sqrt(S)) * lr_mul * W[r]·latent[b,row_lat[r]] + bias[r]*lr_mul  (EqualLinear, model.py:129-158)."""
    lat = _dev(latent, 'latent')
    if lat.ndim == 2:
        lat = lat.unsqueeze(1)
    B, L, S = lat.shape
    w = _dev(wcat, 'wcat')
    R = w.shape[0]
    s = torch.empty(B, R, device=lat.device, dtype=torch.float32)
    scale = (1.0 / math.sqrt(S)) * lr_mul
    check(_lib.lib().oodgan_style_affine_fwd(_p(lat.contiguous()), _p(w), _p(_opt(bcat, 'bcat')), _p(row_lat), _p(s), B, L, S, R,
                                             scale, float(lr_mul), _stream()), 'style_affine_fwd')
    return s


def style_affine_backward(gs, wcat, lat_start, L, lr_mul=1.0, grad_div=1.0):
    g = _dev(gs, 'gs')
    w = _dev(wcat, 'wcat')
    B, R = g.shape
    S = w.shape[1]
    glat = torch.empty(B, L, S, device=g.device, dtype=torch.float32)
    check(_lib.lib().oodgan_style_affine_bwd(_p(g), _p(w), _p(lat_start), _p(glat), B, L, S, R,
                                             (1.0 / math.sqrt(S)) * lr_mul / grad_div, _stream()), 'style_affine_bwd')
    return glat


def equal_linear(x, weight, bias=None, lr_mul=1.0, activation=False):
    x = _dev(x)
    w = _dev(weight, 'weight')
    B, I = x.shape
    O = w.shape[0]
    y = torch.empty(B, O, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_equal_linear(_p(x), _p(w), _p(_opt(bias, 'bias')), _p(y), B, I, O, (1.0 / math.sqrt(I)) * lr_mul,
                                         float(lr_mul), 1 if activation else 0, _stream()), 'equal_linear')
    return y


def equal_linear_grouped(x, weight, bias=None, lr_mul=1.0, activation=False):
    """G EqualLinear layers side by side: x (B,G,I), weight (G,O,I), bias (G,O) -> (B,G,O); bit-identical to G ``equal_linear`` calls."""
    x = _dev(x)
    w = _dev(weight, 'weight')
    B, G, I = x.shape
    O = w.shape[1]
    y = torch.empty(B, G, O, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_equal_linear_grouped(_p(x), _p(w), _p(_opt(bias, 'bias')), _p(y), B, G, I, O, (1.0 / math.sqrt(I)) * lr_mul,
                                                 float(lr_mul), 1 if activation else 0, _stream()), 'equal_linear_grouped')
    return y


def linear(x, weight, bias=None):
    """nn.Linear (feature_style_encoder.py:45,67-68) through the EqualLinear kernel with scale 1."""
    x = _dev(x)
    w = _dev(weight, 'weight')
    B, I = x.shape
    O = w.shape[0]
    y = torch.empty(B, O, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_equal_linear(_p(x), _p(w), _p(_opt(bias, 'bias')), _p(y), B, I, O, 1.0, 1.0, 0, _stream()), 'linear')
    return y


def pixel_norm(x):
    x = _dev(x)
    y = torch.empty_like(x)
    check(_lib.lib().oodgan_pixel_norm(_p(x), _p(y), x.shape[0], x.shape[1], _stream()), 'pixel_norm')
    return y


def weight_sqsum(weight):
    """(Co,Ci,k,k) -> (Co,Ci) sum of squares over the taps."""
    w = _dev(weight, 'weight')
    Co, Ci = w.shape[0], w.shape[1]
    out = torch.empty(Co, Ci, device=w.device, dtype=torch.float32)
    check(_lib.lib().oodgan_weight_sqsum(_p(w), _p(out), Co, Ci, w.shape[2] * w.shape[3], _stream()), 'weight_sqsum')
    return out


def demod(s, wsq, scale):
    s = _dev(s, 's')
    B, Ci = s.shape
    Co = wsq.shape[0]
    d = torch.empty(B, Co, device=s.device, dtype=torch.float32)
    check(_lib.lib().oodgan_demod_fwd(_p(s), Ci, _p(wsq), _p(d), Co, B, Ci, Co, float(scale), _stream()), 'demod_fwd')
    return d


class SForm:
    """S-form activation buffer (csrc/sform.hpp): (B, C, H, W) logical, 64-byte {hi,lo} f16 records per pixel and
    16-channel block, zero border + tile padding.  ``data`` must stay zero outside the interior."""
    __slots__ = ('data', 'B', 'C', 'H', 'W', 'hi_only')

    def __init__(self, B, C, H, W, device):
        n = _lib.lib().oodgan_sform_bytes(B, C, H, W)
        self.data = torch.zeros(n // 2, device=device, dtype=torch.float16)
        self.B, self.C, self.H, self.W = B, C, H, W
        # True: the buffer holds 32-byte hi-only records (oodgan_actbwd_fuse.ys_hi_only) and is DEDICATED to them — its zero border lives at the
        # hi-only addresses; conv3x3 passes x_hi_only = 2
        self.hi_only = False

    def data_ptr(self):
        return self.data.data_ptr()

    @property
    def shape(self):
        return (self.B, self.C, self.H, self.W)


class FForm:
    """fp32 activations in F-form (include/oodgan.h, oodgan_conv_args.y_fform): [B][C/16][H][W][16] — one 64-byte record per pixel
    and 16-channel block — in the storage of a (B,C,H,W) tensor.  Private hand-off between the strip conv of the last styled
    layer and its activation backward inside the W+ loop."""
    __slots__ = ('data', 'B', 'C', 'H', 'W')

    def __init__(self, data):
        self.data = data
        self.B, self.C, self.H, self.W = data.shape

    def data_ptr(self):
        return self.data.data_ptr()

    @property
    def shape(self):
        return (self.B, self.C, self.H, self.W)

    def to_nchw(self):
        y = torch.empty(self.B, self.C, self.H, self.W, device=self.data.device, dtype=torch.float32)
        check(_lib.lib().oodgan_from_fform(_p(self.data), _p(y), self.B, self.C, self.H, self.W, _stream()), 'from_fform')
        return y


class SFormSaved:
    """An activation that exists ONLY as the S-form a conv's epilogue wrote for its consumer (``conv3x3(..., ys=, want_y=False)``): the
    values are x * scale[b,c] split into an f16 pair, ``scale`` (a ``Cols`` / (B,C) tensor: style x range scale of the consumer) is kept
    for the readers that need x itself.  ``shape`` is the logical (B, C, H, W)."""
    __slots__ = ('sform', 'scale')

    def __init__(self, sform, scale):
        self.sform, self.scale = sform, scale

    @property
    def shape(self):
        return self.sform.shape

    def to_nchw(self):
        """x as an fp32 NCHW tensor ((hi + lo) / scale): for the readers that do not take the S-form (the two-pass backward)."""
        B, C, H, W = self.sform.shape
        y = torch.empty(B, C, H, W, device=self.sform.data.device, dtype=torch.float32)
        sc = self.scale
        check(_lib.lib().oodgan_from_sform(_p(self.sform), _p(sc), 0 if sc is None else sc.shape[1], _p(y), B, C, H, W, _stream()), 'from_sform')
        return y


class SFormPhases:
    """Phase-split S-form of a (B, C, 2H+1, 2W+1) tensor (input of the stride-2 conv); H, W = conv output size."""
    __slots__ = ('data', 'B', 'C', 'H', 'W', 'hi_only')

    def __init__(self, B, C, H, W, device):
        n = _lib.lib().oodgan_sform_phases_bytes(B, C, H, W)
        self.data = torch.zeros(n // 2, device=device, dtype=torch.float16)
        self.B, self.C, self.H, self.W = B, C, H, W
        # True while the buffer holds the 32-byte hi-only records of oodgan_act_bwd_blurT_sform_phases_hi (set / cleared by the producers;
        # conv3x3 then passes x_hi_only = 2)
        self.hi_only = False

    def data_ptr(self):
        return self.data.data_ptr()

    @property
    def shape(self):
        return (self.B, self.C, 2 * self.H + 1, 2 * self.W + 1)


def to_sform_phases(x, H, W, scale=None, mul2=None, out=None, in_pitch=0, pad_tl=False):
    """pitched fp32 (B,C,2H+1,pitch) -> phase-split S-form of x*scale[b,c]*mul2[1].  ``pad_tl``: x is the unpadded (B,C,2H,2W) input
    of a stride-2 conv with padding 1; its zero row / column on the top / left are produced by the kernel."""
    x = _dev(x)
    B, C = x.shape[0], x.shape[1]
    if out is None:
        out = SFormPhases(B, C, H, W, x.device)
    out.hi_only = False
    fn = _lib.lib().oodgan_to_sform_phases_padtl if pad_tl else _lib.lib().oodgan_to_sform_phases
    check(fn(_p(x), _p(_opt(scale, 'scale')), 0 if scale is None else scale.shape[1], _p(mul2), _p(out), B, C, H, W, in_pitch, _stream()),
          'to_sform_phases')
    return out


def blurT_to_sform_phases(g, kernel, scale=None, mul2=None, out=None):
    """g (B,C,2H,2W) -> phase-split S-form of upfirdn2d(g, kernel, pad=(2,2)) * scale[b,c] * mul2[1]."""
    g = _dev(g)
    B, C = g.shape[0], g.shape[1]
    H, W = g.shape[2] // 2, g.shape[3] // 2
    if out is None:
        out = SFormPhases(B, C, H, W, g.device)
    out.hi_only = False
    check(_lib.lib().oodgan_blurT_to_sform_phases(_p(g), _p(_dev(kernel)), _p(_opt(scale, 'scale')),
                                                  0 if scale is None else scale.shape[1], _p(mul2), _p(out), B, C, H, W,
                                                  _stream()), 'blurT_to_sform_phases')
    return out


_SFORM_POOL = {}


def sform_phases_scratch(B, C, H, W, device):
    key = ('ph', B, C, H, W, str(device), _stream_handle())
    buf = _SFORM_POOL.get(key)
    if buf is None:
        buf = _SFORM_POOL[key] = SFormPhases(B, C, H, W, device)
    return buf


_WS_POOL = {}
USE_TINY = True      # the 4x4 / 8x8 layers as a skinny GEMM with a K split (csrc/conv_f16s_tiny.hip); False: the tile kernels


def conv_workspace(nbytes, device):
    """Scratch for ``oodgan_conv_args.workspace`` (the K-split partial tiles of csrc/conv_f16s_tiny.hip: written, then read by the
    finishing launch — no counters, no initialisation needed), one per HIP stream — two streams running the same layer at the
    same time must not share partial tiles."""
    key = (nbytes, str(device), _stream_handle())
    buf = _WS_POOL.get(key)
    if buf is None:
        buf = _WS_POOL[key] = torch.empty(nbytes // 4, device=device, dtype=torch.int32)
    return buf


def drop_stream_scratch(stream_handle):
    """Release the pooled S-form buffers and conv workspaces created for one HIP stream (pools are keyed by the raw stream handle:
    an owner that retires its stream — ``GraphedForward.reset`` — must not leave them behind)."""
    for pool in (_SFORM_POOL, _WS_POOL):
        for key in [k for k in pool if k[-1] == stream_handle]:
            del pool[key]
    for key in [k for k in _VMAX_POOL if k[1] == stream_handle]:
        del _VMAX_POOL[key]


def sform_scratch(B, C, H, W, device, tag=0):
    """Reusable S-form buffer (zero border written once at allocation; producers only touch the interior, so a
    buffer can be recycled for any tensor of the same logical shape)."""
    key = (B, C, H, W, str(device), tag, _stream_handle())
    buf = _SFORM_POOL.get(key)
    if buf is None:
        buf = _SFORM_POOL[key] = SForm(B, C, H, W, device)
    return buf


def to_sform(x, scale=None, mul2=None, out=None, in_hw=None, in_pitch=0, vmax=None, shift=None):
    """fp32 NCHW -> S-form of (x*scale[b,c] + shift[b,c])*mul2[1].  ``vmax`` (B int32, zero-initialised float bit patterns):
    per-sample max |value written| for the forward range control (``FwdRange``)."""
    x = _dev(x)
    B, C = x.shape[0], x.shape[1]
    H, W = in_hw if in_hw is not None else (x.shape[2], x.shape[3])
    if out is None:
        out = SForm(B, C, H, W, x.device)
    check(_lib.lib().oodgan_to_sform(_p(x), _p(_opt(scale, 'scale')), 0 if scale is None else scale.shape[1],
                                     _p(_opt(shift, 'shift')), 0 if shift is None else shift.shape[1], _p(mul2), _p(out),
                                     B, C, H, W, in_pitch, _p(vmax), _stream()), 'to_sform')
    return out


def blur_act_sform(z, kernel, H, W, bias=None, noise=None, noise_weight=None, act=True, ys=None, ys_scale=None, vmax=None, rank_one=False):
    """Tail of the up-sampling StyledConv in one pass (include/oodgan.h, oodgan_blur_act_sform / _sform_sep): z (B,C,2H+1,pitch) from
    conv3x3(mode T2) -> y (B,C,2H,2W) and, into ``ys`` (an SForm), y*ys_scale for the next conv.  ``rank_one``: the caller knows
    the kernel to be an outer product (selects the strip-walk kernel)."""
    z = _dev(z)
    B, C = z.shape[0], z.shape[1]
    y = torch.empty(B, C, 2 * H, 2 * W, device=z.device, dtype=torch.float32)
    nz = _opt(noise, 'noise')
    check(_lib.lib().oodgan_blur_act_sform_sep(_p(z), _p(_dev(kernel, 'kernel')), _p(y), _p(ys), _p(_opt(ys_scale, 'ys_scale')),
                                               0 if ys_scale is None else ys_scale.shape[1], _p(_opt(bias, 'bias')), _p(nz),
                                               1 if nz is None else nz.shape[0], _p(_opt(noise_weight, 'nw')),
                                               ACT_LRELU if act else ACT_NONE, B, C, H, W, z.shape[3], _p(vmax), 1 if rank_one else 0,
                                               _stream()), 'blur_act_sform')
    return y


def blur_act_fform(z, kernel, H, W, bias=None, noise=None, noise_weight=None, act=True, ys_scale=None, vmax=None, rank_one=False):
    """The same tail with y in F-form and no S-form (include/oodgan.h, oodgan_blur_act_fform): the following conv converts its input
    itself (``conv3x3(FForm, ..., in_scale=style)``); ``vmax`` still records max |y * ys_scale| for the forward range control.
    ``rank_one``: the caller knows the kernel to be an outer product (selects the strip-walk kernel)."""
    z = _dev(z)
    B, C = z.shape[0], z.shape[1]
    y = torch.empty(B, C, 2 * H, 2 * W, device=z.device, dtype=torch.float32)
    nz = _opt(noise, 'noise')
    check(_lib.lib().oodgan_blur_act_fform(_p(z), _p(_dev(kernel, 'kernel')), _p(y), _p(_opt(ys_scale, 'ys_scale')),
                                           0 if ys_scale is None else ys_scale.shape[1], _p(_opt(bias, 'bias')), _p(nz),
                                           1 if nz is None else nz.shape[0], _p(_opt(noise_weight, 'nw')),
                                           ACT_LRELU if act else ACT_NONE, B, C, H, W, z.shape[3], _p(vmax), 1 if rank_one else 0, _stream()),
          'blur_act_fform')
    return FForm(y)


def xf_supported(B, K, M, H, W):
    """``conv3x3`` takes an ``FForm`` input of this shape (csrc/conv_f16s_stripx.hip)."""
    return bool(_lib.lib().oodgan_conv3x3_xf_supported(B, K, M, H, W))


VMAX_SLOTS = 64      # OODGAN_VMAX_SLOTS (include/oodgan.h)


def absmax_mul2(x):
    """{2^-e, 2^e} (device, 2 floats) with max|x| * 2^e in [512,1024): the power-of-two range scale of a tensor that is
    about to be converted to the S-form (``to_sform(x, mul2=...)``, undone by the consuming conv through ``in_mul2=``).
    e = 0 for an all-zero or non-finite tensor."""
    x = _dev(x)
    B, C = x.shape[0], x.shape[1]
    # float bit patterns, atomic max; ONE persistent slot array per (device, stream, B), zeroed at creation and again by the kernel that
    # reads it (stream order makes the reuse safe; created during warm-up, so never inside a graph capture)
    key = (str(x.device), _stream_handle(), B)
    vm = _VMAX_POOL.get(key)
    if vm is None:
        vm = _VMAX_POOL[key] = torch.zeros(B * VMAX_SLOTS, device=x.device, dtype=torch.int32)
    mul2 = torch.empty(2, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    check(L.oodgan_absmax_scaled(_p(x), None, 0, _p(vm), B, C, x.numel() // (B * C), _stream()), 'absmax_scaled')
    check(L.oodgan_absmax_scale_clear(_p(vm), vm.numel(), _p(mul2), _stream()), 'absmax_scale_clear')
    return mul2


_VMAX_POOL = {}


class FwdRange:
    """Forward range control of the split-f16 path (csrc/fwd_range.hip): one power-of-two scale per styled
    conv and sample, q[l][b], such that the S-form input of conv l holds max|x*s|*q in [512,1024) — an f16 pair cannot
    represent |v| >= 65504 (ModulatedConv2d.forward in fp32, model.py:233-274, has no such limit; trained config-f
    activations reach 1e3..1e4 before modulation).  The producers take the scaled style block ``s_sc`` and the conv
    epilogue the inversely scaled demodulation block ``d_sc``; both are exact power-of-two multiples of the true ones.

    * exact mode (default of a forward): every layer input is measured (``absmax_scaled``) before it is converted;
    * carry mode (steps >= 2 of the W+ loop): the fused producers use the scale measured on the previous step and record
      this step's max; ``finish`` sets ``flag`` when a scaled max left [1, 2^15) or was non-finite (the caller re-runs
      in exact mode) and publishes the next scales."""

    def __init__(self, n_layers, B, R, DR, row_layer, drow_layer, device):
        self.L, self.B, self.R, self.DR = n_layers, B, R, DR
        self.q = torch.ones(n_layers, B, device=device, dtype=torch.float32)
        self.vm = torch.zeros(n_layers, B, VMAX_SLOTS, device=device, dtype=torch.int32)
        self.flag = torch.zeros(1, device=device, dtype=torch.int32)
        self.s_sc = torch.empty(B, R, device=device, dtype=torch.float32)
        self.d_sc = torch.empty(B, DR, device=device, dtype=torch.float32)
        self.row_layer, self.drow_layer = row_layer, drow_layer
        self.valid = False          # q holds measured scales (set by an exact pass)

    def plan(self, s_all, d_all, row0=0, nrows=None, drow0=0, ndrows=None):
        nrows = self.R if nrows is None else nrows
        ndrows = self.DR if ndrows is None else ndrows
        check(_lib.lib().oodgan_fwd_range_plan(_p(s_all), _p(d_all), _p(self.row_layer), _p(self.drow_layer), _p(self.q), _p(self.s_sc),
                                               _p(self.d_sc), self.B, self.R, self.DR, row0, nrows, drow0, ndrows, _stream()),
              'fwd_range_plan')

    def measure(self, l, x, s):
        """exact mode: q[l][:] from max|x*s| (s = the layer's TRUE style block)."""
        x = _dev(x)
        B, C = x.shape[0], x.shape[1]
        check(_lib.lib().oodgan_absmax_scaled(_p(x), _p(s), s.shape[1], _p(self.vm[l]), B, C, x.numel() // (B * C), _stream()),
              'absmax_scaled')
        check(_lib.lib().oodgan_fwd_range_update(_p(self.vm[l]), _p(self.q[l]), None, B, _stream()), 'fwd_range_update')

    def update_exact(self, l):
        """exact mode, when the producer of layer l's input recorded max|x*s| into vm[l] itself (oodgan_upconv_vblur_fform)."""
        check(_lib.lib().oodgan_fwd_range_update(_p(self.vm[l]), _p(self.q[l]), None, self.B, _stream()), 'fwd_range_update')

    def finish(self):
        """carry mode, after the last layer: verify the scales that were used, publish the next ones, clear the maxima."""
        check(_lib.lib().oodgan_fwd_range_update(_p(self.vm), _p(self.q), _p(self.flag), self.L * self.B, _stream()), 'fwd_range_update')

    def violated(self):
        return int(self.flag.item()) != 0


class BwdJobs:
    """Deferred per-layer tail work of one backward pass (reductions of partial sums, demodulation gradient, range-scale
    checks): collected while the gradient chain is enqueued, run as three batched launches at its end.  Keeps the
    tensors alive until then."""

    def __init__(self):
        self.reduce, self.reduce_acc, self.demod, self.check, self.keep = [], [], [], [], []

    def add_reduce(self, part, out_ptr_obj, B, C, nparts, out_stride, accumulate, second=None):
        # accumulating jobs (the dot products) run AFTER the demodulation gradient, as in the per-layer order: the
        # demodulation kernel's `gs += a*b` is one fused multiply-add, so the order of the two contributions is visible
        # in the last bit
        (self.reduce_acc if accumulate else self.reduce).append(_reduce_job(part, out_ptr_obj, B, C, nparts, out_stride, accumulate, second))
        self.keep += [part, out_ptr_obj, second]

    def add_demod(self, s, wsq, d, r, gs, B, Ci, Co, scale):
        self.demod.append(_lib.DemodBwdJob(_p(s), _p(wsq), _p(d), _p(r), _p(gs), s.shape[1], d.shape[1], gs.shape[1], B, Ci, Co,
                                           float(scale)))
        self.keep += [s, wsq, d, r, gs]

    def add_check(self, part, state):
        self.check.append(_lib.ScaleCheckJob(_p(part), part.numel(), _p(state)))
        self.keep += [part, state]

    def run(self, flag):
        L = _lib.lib()
        if self.reduce:
            arr = (_lib.ReduceJob * len(self.reduce))(*self.reduce)
            check(L.oodgan_reduce_batch(arr, len(self.reduce), _stream()), 'reduce_batch')
        if self.demod:
            arr = (_lib.DemodBwdJob * len(self.demod))(*self.demod)
            check(L.oodgan_demod_bwd_batch(arr, len(self.demod), _stream()), 'demod_bwd_batch')
        if self.reduce_acc:
            arr = (_lib.ReduceJob * len(self.reduce_acc))(*self.reduce_acc)
            check(L.oodgan_reduce_batch(arr, len(self.reduce_acc), _stream()), 'reduce_batch')
        if self.check:
            arr = (_lib.ScaleCheckJob * len(self.check))(*self.check)
            check(L.oodgan_absmax_scale_check_batch(arr, len(self.check), _p(flag), _stream()), 'absmax_scale_check_batch')
        self.reduce, self.reduce_acc, self.demod, self.check, self.keep = [], [], [], [], []


def _reduce_job(part, out, B, C, nparts, out_stride, accumulate, second=None):
    """oodgan_reduce_job; ``second`` = (part2 (B,C,nparts2), scale2 (B,*)): out += scale2 * sum part2."""
    j = _lib.ReduceJob(_p(part), _p(out), B, C, nparts, out_stride, 1 if accumulate else 0)
    if second is not None:
        part2, scale2 = second
        j.part2, j.scale2, j.nparts2, j.scale2_stride = _p(part2), _p(scale2), part2.shape[2], scale2.shape[1]
    return j


def _reduce_into(part, B, C, npart, into, accumulate):
    """sum the partials straight into the column block ``into`` (a Cols of the style-gradient accumulator)."""
    check(_lib.lib().oodgan_reduce_parts_cols(_p(part), _p(into), B, C, npart, into.shape[1], 1 if accumulate else 0, _stream()),
          'reduce_parts_cols')


def _reduce_parts(part, rows, npart):
    out = torch.empty(part.shape[0], part.shape[1], device=part.device, dtype=torch.float32)
    check(_lib.lib().oodgan_reduce_parts(_p(part), _p(out), rows, npart, 0, _stream()), 'reduce')
    return out


def blurT_hi_supported(H, W):
    """``act_bwd_producer(..., blur_kernel=k, hi_only=True)`` exists for an up-conv input of H x W."""
    return bool(_lib.lib().oodgan_act_bwd_blurT_hi_supported(H, W))


def act_bwd_producer(out, g_feat, noise, noise_weight, bias, dscale, mul2, dst, g_rgb=None, w_rgb=None, s_rgb=None,
                     blur_kernel=None, t_into=None, jobs=None, dot_of=None, hi_only=False):
    """Fused backward producer (include/oodgan.h): the gradient of bias+noise+lrelu*sqrt2 (+ToRGB branch) of ``out``
    written straight into ``dst`` — an ``SForm`` (plain conv layer) or, with ``blur_kernel``, an ``SFormPhases``
    (up-conv layer: blur^T and phase split fused) — scaled by ``dscale[b,c] * mul2[1]``.
    Returns (r[B,C], t[B,C] or None, part_max) — ``part_max`` goes to ``absmax_scale_check``.
    ``out=None`` (up-conv layers, where ``s1_actgrad_supported``): ``g_feat`` already is g_pre, made by the conv above with
    ``dot_actgrad=dot_of`` (a ``DotActGrad``); the kernel sums the noise / bias term of r and the reduction adds
    out_scale * dot of that conv (include/oodgan.h)."""
    pre = out is None
    assert not pre or (dot_of is not None and dot_of.dot_part is not None)
    fform = isinstance(out, FForm)
    if fform:
        assert g_feat is None and blur_kernel is None, 'F-form: the last styled conv only (no conv gradient, no blur)'
        o = out.data
    else:
        o = _dev(g_feat, 'g_feat') if pre else _dev(out, 'out')
    B, C, H, W = o.shape
    L = _lib.lib()
    up = blur_kernel is not None
    npart = L.oodgan_act_bwd_blurT_nparts(H // 2, W // 2) if up else L.oodgan_act_bwd_sform_nparts(H, W)
    KC = (C + 15) // 16
    part_r = torch.empty(B, C, npart, device=o.device, dtype=torch.float32)
    part_t = torch.empty(B, C, npart, device=o.device, dtype=torch.float32) if g_rgb is not None else None
    part_m = torch.empty(B * KC * npart, device=o.device, dtype=torch.float32)
    nz = _opt(noise, 'noise')
    common = [_p(_opt(g_feat, 'g_feat')), _p(None if pre else o), _p(nz), 1 if nz is None else nz.shape[0], _p(_opt(noise_weight, 'nw')),
              _p(_opt(bias, 'bias')), _p(_opt(g_rgb, 'g_rgb')), _p(None if w_rgb is None else _dev(w_rgb).reshape(3, C)),
              _p(_opt(s_rgb, 's_rgb')), 0 if s_rgb is None else s_rgb.shape[1], 1.0 / math.sqrt(C), _p(_dev(dscale, 'dscale')),
              dscale.shape[1], _p(mul2)]
    if fform:
        check(L.oodgan_act_bwd_sform_f(*common[1:], _p(dst), _p(part_r), _p(part_t), _p(part_m), B, C, H, W, _stream()), 'act_bwd_sform_f')
    elif up and hi_only:
        # 32-byte hi-only records for the two-instruction stride-2 conv (precision 'f16s-g2'): the caller made sure that conv takes them
        assert g_rgb is None and part_t is None
        check(L.oodgan_act_bwd_blurT_sform_phases_hi(*common[:6], *common[11:], _p(_dev(blur_kernel)), _p(dst), _p(part_r), _p(part_m),
                                                     B, C, H // 2, W // 2, _stream()), 'act_bwd_blurT_sform_phases_hi')
        dst.hi_only = True
    elif up:
        check(L.oodgan_act_bwd_blurT_sform_phases(*common, _p(_dev(blur_kernel)), _p(dst), _p(part_r), _p(part_t), _p(part_m),
                                                  B, C, H // 2, W // 2, _stream()), 'act_bwd_blurT_sform_phases')
        dst.hi_only = False
    else:
        check(L.oodgan_act_bwd_sform(*common, _p(dst), _p(part_r), _p(part_t), _p(part_m), B, C, H, W, _stream()),
              'act_bwd_sform')
    second = (dot_of.dot_part, dot_of.scale) if pre else None
    if jobs is not None:            # deferred: the sums are only needed by the batched tail of the backward
        r = torch.empty(B, C, device=o.device, dtype=torch.float32)
        jobs.add_reduce(part_r, r, B, C, npart, C, False, second)
        if part_t is not None:
            jobs.add_reduce(part_t, t_into, B, C, npart, t_into.shape[1], False)
        return r, None, part_m
    if pre:
        r = torch.empty(B, C, device=o.device, dtype=torch.float32)
        job = _reduce_job(part_r, r, B, C, npart, C, False, second)
        check(L.oodgan_reduce_batch(ctypes.byref(job), 1, _stream()), 'reduce_batch')
    else:
        r = _reduce_parts(part_r, B * C, npart)
    t = None
    if part_t is not None:
        if t_into is not None:          # ToRGB style gradient straight into its columns of the accumulator
            _reduce_into(part_t, B, C, npart, t_into, False)
        else:
            t = _reduce_parts(part_t, B * C, npart)
    return r, t, part_m


def absmax_scale_check(part_max, state, flag):
    """state {unscale, scale} (device, 2 floats) is verified against this pass's maxima and replaced by the next scale."""
    check(_lib.lib().oodgan_absmax_scale_check(_p(part_max), part_max.numel(), _p(state), _p(flag), _stream()), 'absmax_scale_check')


class HForm:
    """f16 channel-blocked activations of the fp16 modulated conv (include/oodgan.h, oodgan_modconv_f16)."""
    __slots__ = ('buf', 'B', 'C', 'H', 'W')

    def __init__(self, B, C, H, W, device):
        self.B, self.C, self.H, self.W = B, C, H, W
        self.buf = torch.zeros(_lib.lib().oodgan_hform_bytes(B, C, H, W) // 2, dtype=torch.float16, device=device)

    def data_ptr(self):
        return self.buf.data_ptr()

    def to_nchw(self):
        y = torch.empty(self.B, self.C, self.H, self.W, device=self.buf.device, dtype=torch.float32)
        check(_lib.lib().oodgan_from_hform(_p(self), _p(y), self.B, self.C, self.H, self.W, _stream()), 'from_hform')
        return y


def to_hform(x, out=None):
    """fp32 NCHW -> H-form (f16)."""
    x = _dev(x)
    B, C, H, W = x.shape
    if out is None:
        out = HForm(B, C, H, W, x.device)
    check(_lib.lib().oodgan_to_hform(_p(x), _p(out), B, C, H, W, _stream()), 'to_hform')
    return out


def modconv_f16_pack(weight, style, demodulate=True, act='none', out=None, latent=None, mod_weight=None, mod_bias=None):
    """Per-sample modulated (+demodulated) f16 weights of ModulatedConv2d (model.py:236-241) in MFMA fragment order.
    weight (M,K,3,3) or (1,M,K,3,3) fp32 master, style (B,K) fp32 (already through the modulation EqualLinear) — or
    ``style=None`` with ``latent`` (B,S), ``mod_weight`` (K,S), ``mod_bias`` (K): the EqualLinear runs inside the pack kernel.
    ``out``: a buffer from a previous call with the same shapes (no allocation)."""
    weight = _dev(weight).reshape(weight.shape[-4:])
    M, K = weight.shape[0], weight.shape[1]
    B = (style if style is not None else latent).shape[0]
    wpk = out if out is not None else torch.empty(_lib.lib().oodgan_modconv_f16_wbytes(B, M, K) // 2, dtype=torch.float16, device=weight.device)
    if style is not None:
        style = _dev(style)
        check(_lib.lib().oodgan_modconv_f16_pack(_p(weight), _p(style), style.stride(0), 1.0 / math.sqrt(K * 9), int(demodulate),
                                                 _F16_ACT[act], _p(wpk), B, M, K, _stream()), 'modconv_f16_pack')
    else:
        latent, mod_weight = _dev(latent, 'latent'), _dev(mod_weight, 'mod_weight')
        check(_lib.lib().oodgan_modconv_f16_pack_affine(_p(weight), _p(latent), latent.stride(0), _p(mod_weight), _p(_opt(mod_bias, 'mod_bias')),
                                                        latent.shape[1], 1.0 / math.sqrt(K * 9), int(demodulate), _F16_ACT[act], _p(wpk), B, M, K,
                                                        _stream()), 'modconv_f16_pack_affine')
    return wpk, M, K, act


_F16_ACT = {'none': ACT_NONE, 'lrelu': ACT_LRELU}


def modconv_f16(x, packed, noise=None, noise_w=None, bias=None, out=None):
    """fp16 modulated 3x3 conv + noise + bias + activation on H-form activations (32 -> 32 channels class)."""
    wpk, M, K, act = packed       # the activation is fixed at pack time (its gain is folded into the weights)
    assert isinstance(x, HForm) and x.C == K
    if out is None:
        out = HForm(x.B, M, x.H, x.W, x.buf.device)
    nb = 0
    if noise is not None:
        noise = _dev(noise)
        nb = noise.shape[0]
        assert noise.numel() == nb * x.H * x.W and nb in (1, x.B)
    check(_lib.lib().oodgan_modconv_f16(_p(x), _p(wpk), _p(noise), nb, _p(_opt(noise_w, 'noise_w')), _p(_opt(bias, 'bias')),
                                        _F16_ACT[act], _p(out), x.B, K, M, x.H, x.W, _stream()), 'modconv_f16')
    return out


USE_SFORM = True     # S1 convs of the generator engine take their input through an S-form conversion + LDS-DMA kernel
# default conv arithmetic: 'f16s-g2' (round 6: split-f16 — 3 matrix instructions per product in every forward conv, 2 in the input-gradient convs of
# the W+ loop, where the back-propagated gradient is rounded to f16 before each contraction; identical to 'f16s' for model(x)), 'f16s' (3 everywhere)
# or 'f32' (exact fp32 MFMA).  Evidence for the default: tests/test_hip_wplus_long.py (100-step loss curves vs the reference), test_hip_grad2.py
PRECISION = os.environ.get('OODGAN_PRECISION', 'f16s-g2')


class PackedConv:
    """Packed 3x3 weights for one of the two MFMA conv kernels."""
    __slots__ = ('data', 'unscale', 'precision', 'M', 'x_hi_only')

    def __init__(self, data, unscale, precision, M):
        self.data, self.unscale, self.precision, self.M = data, unscale, precision, M
        # precision 'f16s-g2' (input-gradient packings): the conv drops the lo half of its INPUT operand — g_hi * (w_hi + w_lo), two matrix
        # instructions per product (include/oodgan.h, oodgan_conv_args.x_hi_only)
        self.x_hi_only = False

    def data_ptr(self):
        return self.data.data_ptr()


def pack_conv3x3(weight, scale=1.0, transpose=False, flip=False, precision=None):
    """(Co,Ci,3,3) -> K-major packed weights for the MFMA conv kernels (fp32: wpk[K][9][Mp]; split-f16:
    [K/16][9][hi|lo][2][Mp][8] f16 + power-of-two unscale)."""
    precision = precision or PRECISION
    if precision == 'f16s-g2':       # the packing is the split-f16 one; the input-gradient packings are flagged by the engine
        precision = 'f16s'
    w = _dev(weight, 'weight')
    Co, Ci = w.shape[0], w.shape[1]
    M, K = (Ci, Co) if transpose else (Co, Ci)
    if precision == 'f32':
        Mp = (M + 63) // 64 * 64
        out = torch.empty(K, 9, Mp, device=w.device, dtype=torch.float32)
        check(_lib.lib().oodgan_pack_conv3x3(_p(w), _p(out), Co, Ci, float(scale), int(transpose), int(flip), _stream()), 'pack')
        return PackedConv(out, None, 'f32', M)
    nbytes = _lib.lib().oodgan_pack_conv3x3_f16s_bytes(Co, Ci, int(transpose))
    out = torch.empty(nbytes // 2, device=w.device, dtype=torch.float16)
    unscale = torch.empty(2, device=w.device, dtype=torch.float32)
    check(_lib.lib().oodgan_pack_conv3x3_f16s(_p(w), _p(out), _p(unscale), Co, Ci, float(scale), int(transpose), int(flip),
                                              _stream()), 'pack_f16s')
    return PackedConv(out, unscale, 'f16s', M)


class PackedUpVB:
    """Composite weights of ``upconv_vblur_fform``: two packed 3x3 sets (one per output-row parity) in one buffer."""
    __slots__ = ('data', 'unscale4', 'kh', 'wset_bytes', 'M', 'K')

    def __init__(self, data, unscale4, kh, wset_bytes, M, K):
        self.data, self.unscale4, self.kh, self.wset_bytes, self.M, self.K = data, unscale4, kh, wset_bytes, M, K


def rank_one_factors(k2d):
    """kf = the FLIPPED 4x4 blur kernel as upfirdn2d applies it (src/ops/op/upfirdn2d.py:160-193); returns (kv, kh) with
    kf[a][b] == kv[a] * kh[b], or None if the kernel is not an outer product.  Host values (float64)."""
    kf = torch.flip(k2d.detach().double().cpu(), [0, 1])
    if kf.shape != (4, 4) or not torch.isfinite(kf).all() or float(kf.abs().max()) == 0.0:
        return None
    i, j = divmod(int(kf.abs().argmax()), 4)
    kv, kh = kf[:, j] / kf[i, j], kf[i, :].clone()
    if float((torch.outer(kv, kh) - kf).abs().max()) > 1e-12 * float(kf.abs().max()):
        return None
    return kv, kh


def pack_upconv_vblur(weight, scale, k2d):
    """(Co,Ci,3,3) transposed-conv weight + the 4x4 blur kernel of ``Blur(pad=(1,1))`` -> ``PackedUpVB`` (include/oodgan.h,
    oodgan_upconv_vblur_fform): the vertical blur pass folded into two 3x3 weight sets,
    Wv[py][d+1][kx] = sum_{a,ky: py+a-1-ky == 2d} kv[a] * W[ky][kx]; None if the kernel is not rank one.  One-time weight preparation."""
    f = rank_one_factors(k2d)
    if f is None:
        return None
    kv, kh = f
    w = _dev(weight, 'weight')
    Co, Ci = w.shape[0], w.shape[1]
    wv = torch.zeros(2, Co, Ci, 3, 3, device=w.device, dtype=torch.float64)
    for py in range(2):
        for a in range(4):
            for ky in range(3):
                t = py + a - 1 - ky
                if t % 2 == 0 and -1 <= t // 2 <= 1:
                    wv[py, :, :, t // 2 + 1, :] += float(kv[a]) * w[:, :, ky, :].double()
    wv = wv.float().contiguous()
    L = _lib.lib()
    nbytes = L.oodgan_pack_conv3x3_f16s_bytes(Co, Ci, 0)
    data = torch.empty(nbytes, device=w.device, dtype=torch.float16)          # 2 sets x nbytes / 2 halves
    unscale4 = torch.empty(4, device=w.device, dtype=torch.float32)
    for py in range(2):
        check(L.oodgan_pack_conv3x3_f16s(_p(wv[py]), ctypes.c_void_p(data.data_ptr() + py * nbytes),
                                         ctypes.c_void_p(unscale4.data_ptr() + 8 * py), Co, Ci, float(scale), 0, 0, _stream()), 'pack_f16s')
    return PackedUpVB(data, unscale4, kh.float().to(w.device).contiguous(), nbytes, Co, Ci)


def upconv_vblur_supported(B, K, M, H, W):
    return bool(_lib.lib().oodgan_upconv_vblur_supported(B, K, M, H, W))


def upconv_vblur_fform(xs, wvb, out_scale=None, bias=None, noise=None, noise_weight=None, act=True, ys_scale=None, vmax=None):
    """The up-sampling StyledConv in one pass (include/oodgan.h, oodgan_upconv_vblur_fform): S-form x (B,K,H,W) -> F-form
    act(Blur(conv_transpose2d(x, W)) * out_scale + noise_w * noise + bias) (B,M,2H,2W), without the (2H+1)² intermediate."""
    assert isinstance(xs, SForm) and isinstance(wvb, PackedUpVB)
    B, K, H, W = xs.shape
    assert K == wvb.K
    M = wvb.M
    y = torch.empty(B, M, 2 * H, 2 * W, device=xs.data.device, dtype=torch.float32)
    nz = _opt(noise, 'noise')
    check(_lib.lib().oodgan_upconv_vblur_fform(_p(xs), _p(wvb.data), wvb.wset_bytes, _p(wvb.unscale4), _p(wvb.kh),
                                               _p(_opt(out_scale, 'out_scale')), 0 if out_scale is None else out_scale.shape[1],
                                               _p(_opt(bias, 'bias')), _p(nz), 1 if nz is None else nz.shape[0],
                                               _p(_opt(noise_weight, 'nw')), ACT_LRELU if act else ACT_NONE,
                                               _p(_opt(ys_scale, 'ys_scale')), 0 if ys_scale is None else ys_scale.shape[1], _p(vmax),
                                               _p(y), B, K, M, H, W, _stream()), 'upconv_vblur_fform')
    return FForm(y)


def conv3x3(x, wpk, M, mode=CONV_S1, in_scale=None, in_shift=None, out_scale=None, bias=None, noise=None,
            noise_weight=None, act=ACT_NONE, slope=None, dotx=None, in_hw=None, in_pitch=0, out=None, out_pitch=0,
            in_mul2=None, ys=None, ys_scale=None, want_y=True, dot_into=None, rgb=None, jobs=None, fuse=None, dot_actgrad=None,
            groups=1, y_fform=False, xf_act=None, tiny_max=8, vmax=None):
    """Implicit-GEMM 3x3 conv on the matrix cores.  ``x`` is an fp32 NCHW tensor or an ``SForm`` (split-f16 kernels,
    mode S1).  Returns y, or (y, dot[B,M]) when ``dotx`` is given; ``ys`` (an SForm) additionally receives
    act(y)*ys_scale in S-form for the next conv.  ``groups`` > 1: nn.Conv2d(groups=G) semantics — x has G*K channels, the
    packed weight G*Mg output channels of K inputs (mode S2; fp32 input, or phase-split S-form where ``s2_grouped_supported``).
    ``tiny_max``: largest output size offered to the skinny-GEMM kernel (8: the generator's 4x4 / 8x8 layers; the encoder trunk passes 32,
    which that kernel takes for maps of at most 1024 positions in all)."""
    sform_in = isinstance(x, (SForm, SFormPhases))
    fform_in = isinstance(x, FForm)      # split-f16 strip kernel with in-kernel conversion (include/oodgan.h, x_fform)
    if fform_in:
        assert mode == CONV_S1 and wpk.precision == 'f16s'
        y_fform = xf_act is None        # forward: F-form out; input gradient (xf_act: the activation backward that makes the input): NCHW
    elif not sform_in:
        x = _dev(x)
    B, K = x.shape[0], x.shape[1] // max(1, groups)
    H, W = in_hw if in_hw is not None else (x.shape[2], x.shape[3])
    if mode == CONV_S1:
        oh, ow = H, W
    elif mode == CONV_T2:
        oh, ow = 2 * H + 1, 2 * W + 1
        if out_pitch == 0:
            out_pitch = (ow + 3) // 4 * 4        # 16-byte aligned rows (float4 staging of the consumer), even for float2 stores
    else:
        oh, ow = (H - 1) // 2, (W - 1) // 2
    pitch = out_pitch if out_pitch else ow
    if out is None and want_y:
        out = torch.empty(B, M, oh, pitch, device=x.data.device if (sform_in or fform_in) else x.device, dtype=torch.float32)
    a = ConvArgs()
    a.x, a.wpk, a.y = _p(x), _p(wpk), _p(out)
    a.in_scale, a.in_shift, a.out_scale = _p(_opt(in_scale, 'in_scale')), _p(_opt(in_shift, 'in_shift')), _p(_opt(out_scale, 'out_scale'))
    a.bias, a.noise, a.noise_w, a.slope = _p(_opt(bias, 'bias')), _p(_opt(noise, 'noise')), _p(_opt(noise_weight, 'nw')), _p(_opt(slope, 'slope'))
    a.B, a.K, a.M, a.Hin, a.Win = B, K, M, H, W
    a.in_pitch, a.out_pitch = in_pitch, out_pitch
    a.in_scale_stride = in_scale.shape[1] if in_scale is not None else 0
    a.out_scale_stride = out_scale.shape[1] if out_scale is not None else 0
    a.noise_batch = noise.shape[0] if noise is not None else 1
    a.mode, a.act = mode, act
    a.in_mul2 = _p(in_mul2)
    a.x_sform = 1 if sform_in else 0
    a.x_fform = (2 if xf_act is not None else 1) if fform_in else 0
    ws = None
    if sform_in and USE_TINY and wpk.precision == 'f16s' and mode in (CONV_S1, CONV_S2) and min(oh, ow) <= tiny_max:
        nb = _lib.lib().oodgan_conv3x3_tiny_workspace(mode, B, K, M, H, W)
        if nb > 0:
            ws = conv_workspace(nb, x.data.device)
            a.workspace, a.workspace_bytes = _p(ws), nb
    a.groups = int(groups)
    a.x_hi_only = 1 if (wpk.x_hi_only and dotx is not None) else 0
    if isinstance(x, (SForm, SFormPhases)) and x.hi_only:
        assert a.x_hi_only == 1, 'hi-only input records need the two-instruction input-gradient conv (precision f16s-g2, dotx)'
        a.x_hi_only = 2
    a.y_fform = 1 if y_fform else 0
    a.ys, a.ys_scale = _p(ys), _p(_opt(ys_scale, 'ys_scale'))
    a.ys_vmax = _p(vmax)        # with ys from the 8-wave kernel: max |act(y) * ys_scale| per sample (forward range control of the reader)
    rgb_y = None
    if rgb is not None:         # (w_rgb (3,M), s_rgb Cols/(B,M)): also emit the ToRGB colour sums of the activated output
        w_rgb, s_rgb = rgb
        # the strip kernel (32 channels) writes the complete sums, the 8-wave kernel one partial per 64-channel block: rgb_finish adds them
        nrgb = 1 if M <= 32 else (M + 63) // 64
        rgb_y = torch.empty(*((B, 3, oh, ow) if nrgb == 1 else (nrgb, B, 3, oh, ow)), device=x.data.device if sform_in or fform_in else x.device,
                            dtype=torch.float32)
        a.rgb_w, a.rgb_s, a.rgb_y = _p(_dev(w_rgb, 'w_rgb').reshape(3, M)), _p(_dev(s_rgb, 's_rgb')), _p(rgb_y)
        a.rgb_s_stride, a.rgb_scale = s_rgb.shape[1], 1.0 / math.sqrt(M)
    a.ys_scale_stride = ys_scale.shape[1] if ys_scale is not None else 0
    fz = None
    if fuse is not None:        # ActBwdFusion: the activation backward of the layer below runs in this conv's epilogue
        fz = fuse.struct(B, M, oh, ow)
        a.fuse = ctypes.cast(ctypes.pointer(fz), ctypes.c_void_p)
    if xf_act is not None:      # ActBwdX: the activation backward of THIS layer produces the conv input inside the kernel
        fz = xf_act.struct(B, K, H, W)
        a.fuse = ctypes.cast(ctypes.pointer(fz), ctypes.c_void_p)
    part = None
    if dotx is not None:
        dot_f = isinstance(dotx, FForm)
        dot_s = isinstance(dotx, SFormSaved)      # saved only as its consumer's S-form: the fused stride-2 epilogue decodes it (dotx_sform)
        if dot_s and fuse is None:
            dotx, dot_s = dotx.to_nchw(), False
        dx_ = dotx.data if dot_f else (dotx.sform.data if dot_s else _dev(dotx, 'dotx'))
        a.dotx_fform = 1 if dot_f else 0
        if dot_s:
            a.dotx_sform, a.dotx_scale, a.dotx_scale_stride = 1, _p(dotx.scale), dotx.scale.shape[1]
        if fform_in:
            npart = _lib.lib().oodgan_conv3x3_xf_nparts(B, H, W)
        elif wpk.precision == 'f16s':
            npart = _lib.lib().oodgan_conv3x3_f16s_nparts2(mode, H, W, 1 if sform_in else 0)
        else:
            npart = _lib.lib().oodgan_conv3x3_nparts(mode, H, W)
        part = torch.empty(B, M, npart, device=dx_.device, dtype=torch.float32)
        a.dotx, a.dot_part, a.dot_nparts = _p(dx_), _p(part), npart
        if dot_actgrad is not None:     # DotActGrad: y <- y * act'(dotx); the producer below needs these partials for its r
            a.dot_actgrad = 1
            dot_actgrad.dot_part, dot_actgrad.scale = part, out_scale
    if wpk.precision == 'f16s':
        check(_lib.lib().oodgan_conv3x3_f16s(ctypes.byref(a), _p(wpk.unscale), _stream()), 'conv3x3_f16s')
    else:
        check(_lib.lib().oodgan_conv3x3(ctypes.byref(a), _stream()), 'conv3x3')
    if fuse is not None:
        fuse.finish(part, a.dot_nparts, jobs)
    if xf_act is not None:
        xf_act.finish(jobs)
    if dotx is not None:
        if dot_into is not None:        # += into the layer's columns of the style-gradient accumulator
            if jobs is not None:
                jobs.add_reduce(part, dot_into, B, M, a.dot_nparts, dot_into.shape[1], True)
            else:
                _reduce_into(part, B, M, a.dot_nparts, dot_into, True)
            return out, None
        dot = torch.empty(B, M, device=dx_.device, dtype=torch.float32)
        check(_lib.lib().oodgan_reduce_parts(_p(part), _p(dot), B * M, a.dot_nparts, 0, _stream()), 'reduce_parts')
        return out, dot
    if y_fform:
        out = FForm(out)
    if rgb is not None:
        return out, rgb_y
    return out


def s2_fuse_supported(B, K, M, Hin, Win):
    return bool(_lib.lib().oodgan_conv3x3_s2_fuse_supported(B, K, M, Hin, Win))


def tiny_workspace_bytes(mode, B, K, M, Hin, Win):
    """> 0 when the skinny-GEMM kernel (outputs of 8 x 8 and below) takes an S-form input of this shape; K per group."""
    return int(_lib.lib().oodgan_conv3x3_tiny_workspace(mode, B, K, M, Hin, Win)) if USE_TINY else 0


def s1_xh_supported(B, K, M, H, W):
    """mode S1 with an S-form input and ``dotx`` runs the 8-wave kernel, whose two-instruction instances read 32-byte hi-only records."""
    return bool(_lib.lib().oodgan_conv3x3_s1_xh_supported(B, K, M, H, W))


def sform_hi_scratch(B, C, H, W, device):
    """Reusable S-form buffer DEDICATED to 32-byte hi-only records (its zero border lives at their addresses)."""
    buf = sform_scratch(B, C, H, W, device, tag='hi')
    buf.hi_only = True
    return buf


def s1_ys_supported(B, K, M, H, W):
    """mode S1 with an S-form input writes ``ys`` from the 8-wave kernel (``want_y=False``: nothing but the S-form is produced)."""
    return bool(_lib.lib().oodgan_conv3x3_s1_ys_supported(B, K, M, H, W))


def s2_grouped_supported(B, K, M, groups, Hin, Win):
    """mode S2 with ``groups`` > 1 takes a phase-split S-form input (K channels per group, M = groups * Mg outputs)."""
    return bool(_lib.lib().oodgan_conv3x3_s2_grouped_supported(B, K, M, groups, Hin, Win))


def s1_actgrad_supported(B, K, M, H, W):
    """mode S1 with an S-form input and ``dotx`` takes ``dot_actgrad`` and the blur^T producer below takes ``out=None``."""
    return bool(_lib.lib().oodgan_conv3x3_s1_actgrad_supported(B, K, M, H, W))


class DotActGrad:
    """Link between ``conv3x3(..., mode=CONV_S1, dotx=out_below, dot_actgrad=this)`` — the stride-1 input-gradient conv
    above an up-sampling layer, which then returns g_pre = dx * act'(out_below) instead of dx — and
    ``act_bwd_producer(None, g_pre, ..., blur_kernel=k, dot_of=this)``, which finishes that layer's activation backward
    from g_pre alone: carries the conv's dot partials and out_scale, the ``sum dx*out`` term of the layer's r."""
    __slots__ = ('dot_part', 'scale')

    def __init__(self):
        self.dot_part = self.scale = None


class ActBwdFusion:
    """Arguments and results of the activation backward fused into the stride-2 input-gradient conv
    (oodgan_actbwd_fuse, include/oodgan.h): the same quantities ``act_bwd_producer`` returns — (r, t, part_max) and the
    S-form gradient ``dst`` — produced by ``conv3x3(..., mode=CONV_S2, fuse=this)`` for the layer whose output is ``dotx``."""

    def __init__(self, dst, noise, noise_weight, bias, dscale, mul2, g_rgb=None, w_rgb=None, s_rgb=None, t_into=None, hi_only=False):
        self.hi_only = bool(hi_only)        # dst receives 32-byte hi-only records (a buffer dedicated to them: sform_scratch(tag='hi'))
        self.dst, self.noise, self.noise_weight, self.bias, self.dscale, self.mul2 = dst, _opt(noise, 'noise'), _opt(noise_weight, 'nw'), _opt(bias, 'bias'), dscale, mul2
        self.g_rgb, self.w_rgb, self.s_rgb, self.t_into = _opt(g_rgb, 'g_rgb'), w_rgb, s_rgb, t_into
        self.r = self.t = self.part_m = None

    def struct(self, B, M, H, W):
        dev = self.dst.data.device
        ntile = ((H + 7) // 8) * ((W + 31) // 32)
        self.B, self.M = B, M
        self.part_r = torch.empty(B, M, ntile, device=dev, dtype=torch.float32)
        self.part_t = torch.empty(B, M, ntile, device=dev, dtype=torch.float32) if self.g_rgb is not None else None
        self.part_m = zeros(B * ntile * ((M + 63) // 64) * 8, device=dev)
        z = _lib.ActBwdFuse()
        z.g_rgb, z.noise, z.noise_w, z.bias = _p(self.g_rgb), _p(self.noise), _p(self.noise_weight), _p(self.bias)
        self._w = None if self.w_rgb is None else _dev(self.w_rgb).reshape(3, M)
        z.w_rgb, z.s_rgb = _p(self._w), _p(self.s_rgb)
        z.s_rgb_stride = 0 if self.s_rgb is None else self.s_rgb.shape[1]
        z.noise_batch = 1 if self.noise is None else self.noise.shape[0]
        z.dscale, z.dscale_stride, z.mul2, z.ys = _p(self.dscale), self.dscale.shape[1], _p(self.mul2), _p(self.dst)
        z.part_r, z.part_t, z.part_max = _p(self.part_r), _p(self.part_t), _p(self.part_m)
        z.rgb_scale, z.nmax = 1.0 / math.sqrt(M), self.part_m.numel()
        z.ys_hi_only = 1 if self.hi_only else 0
        assert self.dst.hi_only == self.hi_only, 'a hi-only S-form scratch is dedicated to hi-only records'
        return z

    def finish(self, dot_part, nparts, jobs):
        B, M = self.B, self.M
        if jobs is not None:            # deferred, as in act_bwd_producer
            self.r = torch.empty(B, M, device=self.part_r.device, dtype=torch.float32)
            jobs.add_reduce(self.part_r, self.r, B, M, nparts, M, False)
            if self.part_t is not None:
                jobs.add_reduce(self.part_t, self.t_into, B, M, nparts, self.t_into.shape[1], False)
            return
        self.r = _reduce_parts(self.part_r, B * M, nparts)
        if self.part_t is not None:
            if self.t_into is not None:
                _reduce_into(self.part_t, B, M, nparts, self.t_into, False)
            else:
                self.t = _reduce_parts(self.part_t, B * M, nparts)


class ActBwdX:
    """Arguments and results of the activation backward that runs INSIDE the input-gradient conv of the same layer
    (``conv3x3(out_fform, wpk_bwd, ..., xf_act=this)``, oodgan_conv_args.x_fform = 2): the quantities ``act_bwd_producer`` returns
    — (r, t, part_max) — without the S-form gradient ever going to HBM.  Last styled conv only: its gradient comes from ToRGB alone."""

    def __init__(self, noise, noise_weight, bias, dscale, mul2, g_rgb, w_rgb, s_rgb, t_into=None):
        self.noise, self.noise_weight, self.bias, self.dscale, self.mul2 = _opt(noise, 'noise'), _opt(noise_weight, 'nw'), _opt(bias, 'bias'), dscale, mul2
        self.g_rgb, self.w_rgb, self.s_rgb, self.t_into = _dev(g_rgb, 'g_rgb'), w_rgb, s_rgb, t_into
        self.r = self.t = self.part_m = None

    def struct(self, B, C, H, W):
        dev = self.g_rgb.device
        self.B, self.C = B, C
        self.nparts = _lib.lib().oodgan_conv3x3_xf_nparts(B, H, W)
        self.part_r = torch.empty(B, C, self.nparts, device=dev, dtype=torch.float32)
        self.part_t = torch.empty(B, C, self.nparts, device=dev, dtype=torch.float32)
        self.part_m = torch.empty(B * ((C + 15) // 16) * self.nparts, device=dev, dtype=torch.float32)
        z = _lib.ActBwdFuse()
        z.g_rgb, z.noise, z.noise_w, z.bias = _p(self.g_rgb), _p(self.noise), _p(self.noise_weight), _p(self.bias)
        self._w = _dev(self.w_rgb).reshape(3, C)
        z.w_rgb, z.s_rgb, z.s_rgb_stride = _p(self._w), _p(self.s_rgb), self.s_rgb.shape[1]
        z.noise_batch = 1 if self.noise is None else self.noise.shape[0]
        z.dscale, z.dscale_stride, z.mul2, z.ys = _p(self.dscale), self.dscale.shape[1], _p(self.mul2), None
        z.part_r, z.part_t, z.part_max = _p(self.part_r), _p(self.part_t), _p(self.part_m)
        z.rgb_scale, z.nmax = 1.0 / math.sqrt(C), self.part_m.numel()
        return z

    def finish(self, jobs):
        B, C = self.B, self.C
        if jobs is not None:
            self.r = torch.empty(B, C, device=self.part_r.device, dtype=torch.float32)
            jobs.add_reduce(self.part_r, self.r, B, C, self.nparts, C, False)
            if self.t_into is not None:
                jobs.add_reduce(self.part_t, self.t_into, B, C, self.nparts, self.t_into.shape[1], False)
            else:
                self.t = torch.empty(B, C, device=self.part_r.device, dtype=torch.float32)
                jobs.add_reduce(self.part_t, self.t, B, C, self.nparts, C, False)
            return
        self.r = _reduce_parts(self.part_r, B * C, self.nparts)
        if self.t_into is not None:
            _reduce_into(self.part_t, B, C, self.nparts, self.t_into, False)
        else:
            self.t = _reduce_parts(self.part_t, B * C, self.nparts)


def rgb_finish(partial, bias=None, skip=None, kernel=None):
    """second half of ToRGB.forward (model.py:363-372) for colour sums produced by conv3x3(rgb=...): + bias + upsampled skip
    (in place)."""
    if partial.dim() == 5:      # (nparts, B, 3, H, W): partial sums of the 8-wave kernel's 64-channel blocks, added in order into part 0
        n, B, _, H, W = partial.shape
        check(_lib.lib().oodgan_rgb_finish_parts(_p(partial), n, _p(_opt(bias, 'bias')), _p(_opt(skip, 'skip')), _p(_opt(kernel, 'kernel')),
                                                 _p(partial), B, H, W, _stream()), 'rgb_finish_parts')
        return partial[0]
    B, _, H, W = partial.shape
    check(_lib.lib().oodgan_rgb_finish(_p(partial), _p(_opt(bias, 'bias')), _p(_opt(skip, 'skip')), _p(_opt(kernel, 'kernel')),
                                       _p(partial), B, H, W, _stream()), 'rgb_finish')
    return partial


def torgb(x, weight, s, bias=None, skip=None, kernel=None, ys=None, ys_scale=None, vmax=None):
    """ToRGB.forward (src/ops/StyleGAN/model.py:363-372): weight (3,Ci), s (B,Ci) style, skip (B,3,H/2,W/2).
    ``ys`` (an SForm of x's shape): additionally receives x*ys_scale in S-form for the up-sampling conv that follows."""
    x = _dev(x)
    B, Ci, H, W = x.shape
    y = torch.empty(B, 3, H, W, device=x.device, dtype=torch.float32)
    w = _dev(weight, 'weight').reshape(3, Ci)
    if ys is not None:
        check(_lib.lib().oodgan_torgb_fwd_sform(_p(x), _p(w), _p(_dev(s, 's')), s.shape[1], _p(_opt(bias, 'bias')), _p(_opt(skip, 'skip')),
                                                _p(_opt(kernel, 'kernel')), _p(y), _p(ys), _p(_opt(ys_scale, 'ys_scale')),
                                                0 if ys_scale is None else ys_scale.shape[1], B, Ci, H, W, 1.0 / math.sqrt(Ci),
                                                _p(vmax), _stream()), 'torgb_fwd_sform')
        return y
    check(_lib.lib().oodgan_torgb_fwd(_p(x), _p(w), _p(_dev(s, 's')), s.shape[1], _p(_opt(bias, 'bias')), _p(_opt(skip, 'skip')),
                                      _p(_opt(kernel, 'kernel')), _p(y), B, Ci, H, W, 1.0 / math.sqrt(Ci), _stream()), 'torgb_fwd')
    return y


def act_bwd_fused(out, g_feat=None, noise=None, noise_weight=None, bias=None, g_rgb=None, w_rgb=None, s_rgb=None,
                  want_scale=False, dscale=None):
    """Backward through bias+noise+lrelu*sqrt2 merged with the ToRGB branch; returns
    (g_pre, r[B,C] = sum g_pre*y_cv, t[B,C] = sum out*t or None[, mul2]) where mul2 = device {2^-e, 2^e} with
    max|g_pre*dscale|*2^e in [512,1024) (range control for the split-f16 convs; ``dscale`` = the demodulation block the
    consumer multiplies g_pre with before the f16 split)."""
    o = _dev(out, 'out')
    B, C = o.shape[0], o.shape[1]
    HW = o.numel() // (B * C)
    L = _lib.lib()
    npart = L.oodgan_act_bwd_nparts(HW)
    g_pre = torch.empty_like(o)
    part_r = torch.empty(B, C, npart, device=o.device, dtype=torch.float32)
    part_t = torch.empty(B, C, npart, device=o.device, dtype=torch.float32) if g_rgb is not None else None
    nz = _opt(noise, 'noise')
    part_m = torch.empty(B, C, npart, device=o.device, dtype=torch.float32) if want_scale else None
    check(L.oodgan_act_bwd_fused_max(_p(_opt(g_feat, 'g_feat')), _p(o), _p(nz), 1 if nz is None else nz.shape[0],
                                     _p(_opt(noise_weight, 'nw')), _p(_opt(bias, 'bias')), _p(_opt(g_rgb, 'g_rgb')),
                                     _p(None if w_rgb is None else _dev(w_rgb).reshape(3, C)), _p(_opt(s_rgb, 's_rgb')),
                                     0 if s_rgb is None else s_rgb.shape[1], 1.0 / math.sqrt(C), _p(g_pre), _p(part_r),
                                     _p(part_t), _p(part_m), _p(_opt(dscale, 'dscale')), 0 if dscale is None else dscale.shape[1],
                                     B, C, HW, _stream()), 'act_bwd_fused')
    r = torch.empty(B, C, device=o.device, dtype=torch.float32)
    check(L.oodgan_reduce_parts(_p(part_r), _p(r), B * C, npart, 0, _stream()), 'reduce')
    t = None
    if part_t is not None:
        t = torch.empty(B, C, device=o.device, dtype=torch.float32)
        check(L.oodgan_reduce_parts(_p(part_t), _p(t), B * C, npart, 0, _stream()), 'reduce')
    if want_scale:
        mul2 = torch.empty(2, device=o.device, dtype=torch.float32)
        check(L.oodgan_absmax_scale(_p(part_m), part_m.numel(), _p(mul2), _stream()), 'absmax_scale')
        return g_pre, r, t, mul2
    return g_pre, r, t


def demod_backward(s, wsq, d, r, gs, scale):
    """gs += d(demod)/d(style) contribution (in place)."""
    B, Ci = s.shape
    Co = wsq.shape[0]
    check(_lib.lib().oodgan_demod_bwd(_p(_dev(s)), Ci, _p(wsq), _p(_dev(d)), Co, _p(_dev(r)), _p(gs), gs.shape[1], B, Ci, Co,
                                      float(scale), _stream()), 'demod_bwd')
    return gs


def loss_scale_for(chw):
    """Power of two ~ CHW/2 so that grad_mul*2*(img-target)/CHW is O(img-target)."""
    return float(2 ** max(0, int(math.floor(math.log2(max(chw, 2) / 2.0)))))


def mse_loss_grad(img, target, grad_mul=1.0, loss_out=None, table=None, row_dev=None):
    """Per-image MSE and its gradient (times grad_mul): (loss[B], gimg).  ``loss_out``: a contiguous float32 (B,) view that receives the losses
    (the W+ loop's row of its loss table: no copy kernel per step).  ``table`` (nrows, B) + ``row_dev`` (int32[1] on the device): the losses go
    to row row_dev[0] of the table (oodgan_mse_fwd_bwd_row: a recorded step writes a new row on every replay); returns (None, gimg)."""
    a, t = _dev(img, 'img'), _dev(target, 'target')
    B = a.shape[0]
    CHW = a.numel() // B
    L = _lib.lib()
    npart = L.oodgan_mse_nparts(CHW)
    part = torch.empty(B, npart, device=a.device, dtype=torch.float32)
    if table is not None:
        assert table.dim() == 2 and table.shape[1] == B and table.dtype == torch.float32 and table.is_contiguous() and row_dev.dtype == torch.int32
        g = torch.empty_like(a)
        check(L.oodgan_mse_fwd_bwd_row(_p(a), _p(t), _p(g), _p(part), _p(table), _p(row_dev), table.shape[0], B, CHW, float(grad_mul), _stream()), 'mse_row')
        return None, g
    loss = torch.empty(B, device=a.device, dtype=torch.float32) if loss_out is None else loss_out
    assert loss.shape == (B,) and loss.dtype == torch.float32 and loss.is_contiguous() and loss.device == a.device
    g = torch.empty_like(a)
    check(L.oodgan_mse_fwd_bwd(_p(a), _p(t), _p(g), _p(part), _p(loss), B, CHW, float(grad_mul), _stream()), 'mse')
    return loss, g


class LaunchPlan:
    """oodgan_plan_* (include/oodgan.h): the kernel launches this thread makes inside ``with plan.recording():`` — through any op of this
    module — recorded once and re-issued from C++ by ``run()``; eager launches on the streams they were recorded with.  The caller keeps
    every buffer the recorded calls touched allocated at the same address (``WPlusInverter`` records under a private allocator pool)."""

    def __init__(self):
        self._h = _lib.lib().oodgan_plan_create()
        if not self._h:
            raise RuntimeError('oodgan_plan_create failed')
        self.size = 0

    class _Rec:
        def __init__(self, plan):
            self.plan = plan

        def __enter__(self):
            check(_lib.lib().oodgan_plan_record_begin(self.plan._h), 'plan_record_begin')
            return self.plan

        def __exit__(self, *exc):
            n = _lib.lib().oodgan_plan_record_end(self.plan._h)
            if n < 0:
                check(1, 'plan_record_end')
            self.plan.size = int(n)
            return False

    def recording(self):
        return LaunchPlan._Rec(self)

    def run(self, times=1):
        check(_lib.lib().oodgan_plan_run(self._h, int(times)), 'plan_run')

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.lib().oodgan_plan_destroy(h)
            except Exception:
                pass


def adam_step(w, g, m, v, step, lr=0.01, betas=(0.9, 0.999), eps=1e-8):
    check(_lib.lib().oodgan_adam_step(_p(w), _p(_dev(g)), _p(m), _p(v), w.numel(), float(lr), float(betas[0]), float(betas[1]),
                                      float(eps), int(step), _stream()), 'adam')
    return w


def adam_step_dev(w, g, m, v, t_dev, lr=0.01, betas=(0.9, 0.999), eps=1e-8):
    """Adam with the step counter on the device (t_dev int32[1] is incremented first): graph-replayable."""
    check(_lib.lib().oodgan_adam_step_dev(_p(w), _p(_dev(g)), _p(m), _p(v), w.numel(), float(lr), float(betas[0]),
                                          float(betas[1]), float(eps), _p(t_dev), _stream()), 'adam_dev')
    return w
