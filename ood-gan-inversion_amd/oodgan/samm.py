"""SAMM / SAIM — spatial alignment + invertibility mask — on the HIP ops.

Mirrors reference src/ops/SAMM/helpers.py (AlignNet :85-109, SPM_Warp :111-179,
StyledscaleNshfitBlock :182-216, new_PRM :62-77) and bottleneck_IR
(src/ops/e4e/encoders/helpers.py:426-448).  torch.nn containers are used only to hold parameters
under the reference's state-dict keys; every forward runs HIP kernels."""
import ctypes
import math
import os
from ctypes import POINTER, c_float, c_int, c_long, c_void_p

import torch
from torch import nn

from . import _lib, ops
from ._lib import ACT_NONE, ACT_PRELU, CONV_S1, check
from .ops import _dev, _opt, _p, _stream
from .synth import make_kernel

P = c_void_p
FUSE_STATS = int(os.environ.get('OODGAN_SAMM_FUSE_STATS', '1'))     # statistics of a bottleneck's output computed by the pass that writes it
FUSE_CONV_CHAIN = int(os.environ.get('OODGAN_SAMM_FUSE_CHAIN', '1'))     # AlignNet conv -> PReLU -> conv without the fp32 tensor in between
_lib.bind_extra({
    'oodgan_instnorm_stats': (c_int, [P, P, c_int, c_int, c_long, c_float, P]),
    'oodgan_instnorm_coeffs': (c_int, [P, P, P, P, P, c_int, c_int, P]),
    'oodgan_affine_apply': (c_int, [P, P, P, P, P, c_int, c_int, c_long, P]),
    'oodgan_affine_apply_stats': (c_int, [P, P, P, P, P, P, c_int, c_int, c_long, c_float, P]),
    'oodgan_align_input': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_long, P]),
    'oodgan_align_input_stats': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_long, c_float, P]),
    'oodgan_conv1x1': (c_int, [P, P, P, P, c_int, c_int, c_int, c_long, P]),
    'oodgan_conv3x3_small': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_se_gate': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'oodgan_conv3x3_fewout_ksplit': (c_int, [c_int, c_int, c_int, c_int]),
    'oodgan_conv3x3_fewout': (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_conv3x3_fewout2': (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_align_head': (c_int, [P, P, c_int, c_long, c_float, P]),
    'oodgan_field_compose': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_int, P]),
    'oodgan_warp_blend': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    'oodgan_mask_blend': (c_int, [POINTER(c_void_p), POINTER(c_int), c_int, P, P, P, P, c_int, c_int, P]),
    'oodgan_resize_nearest': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'oodgan_resize_bilinear': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
})


# ----------------------------------------------------------------------------- functional layer
def instnorm_stats(x, eps=1e-5):
    x = _dev(x)
    B, C = x.shape[0], x.shape[1]
    st = torch.empty(B, C, 2, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_instnorm_stats(_p(x), _p(st), B, C, x.numel() // (B * C), eps, _stream()), 'instnorm_stats')
    return st


def instnorm_coeffs(stats, gamma=None, beta=None):
    B, C = stats.shape[0], stats.shape[1]
    sc = torch.empty(B, C, device=stats.device, dtype=torch.float32)
    sh = torch.empty_like(sc)
    check(_lib.lib().oodgan_instnorm_coeffs(_p(stats), _p(_opt(gamma, 'gamma')), _p(_opt(beta, 'beta')), _p(sc), _p(sh), B, C,
                                            _stream()), 'instnorm_coeffs')
    return sc, sh


def affine_apply(x, sc, sh, res=None):
    x = _dev(x)
    B, C = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    check(_lib.lib().oodgan_affine_apply(_p(x), _p(sc), _p(sh), _p(_opt(res, 'res')), _p(y), B, C, x.numel() // (B * C),
                                         _stream()), 'affine_apply')
    return y


def affine_apply_stats(x, sc, sh, res=None, eps=1e-5):
    """(affine_apply(x, sc, sh, res), instnorm_stats of it) in one pass over the tensors."""
    x = _dev(x)
    B, C = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    st = torch.empty(B, C, 2, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_affine_apply_stats(_p(x), _p(sc), _p(sh), _p(_opt(res, 'res')), _p(y), _p(st), B, C, x.numel() // (B * C), eps,
                                               _stream()), 'affine_apply_stats')
    return y, st


def instance_norm(x, gamma=None, beta=None, eps=1e-5, res=None, want_stats=False):
    sc, sh = instnorm_coeffs(instnorm_stats(x, eps), gamma, beta)
    if want_stats:
        return affine_apply_stats(x, sc, sh, res, eps)
    return affine_apply(x, sc, sh, res)


def align_input(gen, enc, st_gen, st_enc, diff=True):
    gen, enc = _dev(gen), _dev(enc)
    B, C, H, W = gen.shape
    out = torch.empty(B, 2 * C, H, W, device=gen.device, dtype=torch.float32)
    check(_lib.lib().oodgan_align_input(_p(gen), _p(enc), _p(st_gen), _p(st_enc), _p(out), 1 if diff else 0, B, C, H * W, _stream()), 'align_input')
    return out


def align_input_stats(gen, enc, st_gen, st_enc, eps=1e-5, diff=True):
    """(align_input(...), instnorm_stats of it) in one pass."""
    gen, enc = _dev(gen), _dev(enc)
    B, C, H, W = gen.shape
    out = torch.empty(B, 2 * C, H, W, device=gen.device, dtype=torch.float32)
    st = torch.empty(B, 2 * C, 2, device=gen.device, dtype=torch.float32)
    check(_lib.lib().oodgan_align_input_stats(_p(gen), _p(enc), _p(st_gen), _p(st_enc), _p(out), _p(st), 1 if diff else 0, B, C, H * W, eps, _stream()),
          'align_input_stats')
    return out, st


def conv1x1(x, weight, bias=None):
    x = _dev(x)
    B, K, H, W = x.shape
    w = _dev(weight).reshape(weight.shape[0], K)
    M = w.shape[0]
    y = torch.empty(B, M, H, W, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_conv1x1(_p(x), _p(w), _p(_opt(bias, 'bias')), _p(y), B, K, M, H * W, _stream()), 'conv1x1')
    return y


def se_gate(stats, w1, w2):
    """sigmoid(fc2(relu(fc1(mean)))) of SEModule (e4e/encoders/helpers.py:60-76); stats from ``instnorm_stats``; -> (B, C)."""
    B, C = stats.shape[0], stats.shape[1]
    w1, w2 = _dev(w1).reshape(-1, C), _dev(w2).reshape(C, -1)
    g = torch.empty(B, C, device=stats.device, dtype=torch.float32)
    check(_lib.lib().oodgan_se_gate(_p(stats), _p(w1), _p(w2), _p(g), B, C, w1.shape[0], _stream()), 'se_gate')
    return g


def conv3x3_small(x, weight, in_sc=None, in_sh=None, slope=None):
    x = _dev(x)
    B, K, H, W = x.shape
    w = _dev(weight)
    M = w.shape[0]
    y = torch.empty(B, M, H, W, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_conv3x3_small(_p(x), _p(w), _p(in_sc), _p(in_sh), _p(_opt(slope, 'slope')), _p(y), B, K, M, H, W,
                                          _stream()), 'conv3x3_small')
    return y


def conv3x3_fewout(x, weight, in_sc=None, in_sh=None, slope=None):
    """3x3 conv (pad 1) from many channels to M <= 4, exact fp32, K split over workgroups (include/oodgan.h)."""
    x = _dev(x)
    B, K, H, W = x.shape
    w = _dev(weight)
    M = w.shape[0]
    L = _lib.lib()
    ks = L.oodgan_conv3x3_fewout_ksplit(B, K, H, W)
    part = torch.empty(B, ks, M, H, W, device=x.device, dtype=torch.float32)
    y = torch.empty(B, M, H, W, device=x.device, dtype=torch.float32)
    check(L.oodgan_conv3x3_fewout(_p(x), _p(w), _p(in_sc), _p(in_sh), _p(_opt(slope, 'slope')), _p(part), _p(y), B, K, M, H, W,
                                  _stream()), 'conv3x3_fewout')
    return y


def fewout_weights(w, w11=None):
    """(K, 9, 4) / (K, 4) transposed, zero-filled copies of a (M <= 4, K, 3, 3) weight and a (M2 <= 4, K[,1,1]) 1x1 weight."""
    w = _dev(w)
    M, K = w.shape[0], w.shape[1]
    wt = torch.zeros(K, 9, 4, device=w.device, dtype=torch.float32)
    wt[:, :, :M] = w.reshape(M, K, 9).permute(1, 2, 0)
    w11t = None
    if w11 is not None:
        w11 = _dev(w11).reshape(w11.shape[0], K)
        w11t = torch.zeros(K, 4, device=w.device, dtype=torch.float32)
        w11t[:, :w11.shape[0]] = w11.t()
    return wt, w11t


def conv3x3_fewout2(x, wt, M, in_sc=None, in_sh=None, slope=None, w11t=None, M2=0):
    """conv3x3_fewout from prepared weights (``fewout_weights``); with ``w11t`` also the 1x1 conv of the raw x: -> (y, y2 | None)."""
    x = _dev(x)
    B, K, H, W = x.shape
    L = _lib.lib()
    ks = L.oodgan_conv3x3_fewout_ksplit(B, K, H, W)
    part = torch.empty(B, ks, M, H, W, device=x.device, dtype=torch.float32)
    y = torch.empty(B, M, H, W, device=x.device, dtype=torch.float32)
    part2 = y2 = None
    if w11t is not None:
        part2 = torch.empty(B, ks, M2, H, W, device=x.device, dtype=torch.float32)
        y2 = torch.empty(B, M2, H, W, device=x.device, dtype=torch.float32)
    check(L.oodgan_conv3x3_fewout2(_p(x), _p(wt), _p(w11t), _p(in_sc), _p(in_sh), _p(_opt(slope, 'slope')), _p(part), _p(part2), _p(y), _p(y2),
                                   B, K, M, M2, H, W, _stream()), 'conv3x3_fewout2')
    return y, y2


def align_head(x, scale):
    x = _dev(x)
    y = torch.empty_like(x)
    check(_lib.lib().oodgan_align_head(_p(x), _p(y), x.shape[0], x.shape[2] * x.shape[3], float(scale), _stream()), 'align_head')
    return y


def field_add(acc, cur, scale):
    """SPM_Warp.add (helpers.py:129-137)."""
    acc, cur = _dev(acc), _dev(cur)
    B, _, H, W = cur.shape
    out = torch.empty_like(cur)
    check(_lib.lib().oodgan_field_compose(_p(acc), _p(cur), None, _p(out), B, H, W, 0, 0, float(scale), 0, _stream()), 'field_add')
    return out


def field_upsample_add(prev, cur, scale=0.0):
    """SPM_Warp.upsample_add (helpers.py:139-147): alpha of the coarser level is bicubic(align_corners) upsampled."""
    prev, cur = _dev(prev), _dev(cur)
    B, _, H, W = cur.shape
    out = torch.empty_like(cur)
    check(_lib.lib().oodgan_field_compose(None, _p(cur), _p(prev), _p(out), B, H, W, prev.shape[2], prev.shape[3], float(scale), 1,
                                          _stream()), 'field_upsample_add')
    return out


def warp_blend(target, field):
    target, field = _dev(target), _dev(field)
    B, C, H, W = target.shape
    y = torch.empty_like(target)
    check(_lib.lib().oodgan_warp_blend(_p(target), _p(field), _p(y), B, C, H, W, _stream()), 'warp_blend')
    return y


def mask_blend(fields, x=None, gen=None, size=1024):
    """blending_mask + blend (OOD_faceGAN_e4e_arch.py:315-347).  fields: list of (B,3,s,s) coarse->fine.
    Returns (alpha (B,1,S,S), out or None)."""
    fields = [_dev(f) for f in fields]
    B = fields[0].shape[0]
    n = len(fields)
    ptrs = (c_void_p * n)(*[f.data_ptr() for f in fields])
    sizes = (c_int * n)(*[f.shape[-1] for f in fields])
    alpha = torch.empty(B, 1, size, size, device=fields[0].device, dtype=torch.float32)
    out = torch.empty(B, 3, size, size, device=alpha.device, dtype=torch.float32) if gen is not None else None
    check(_lib.lib().oodgan_mask_blend(ptrs, sizes, n, _p(_opt(x, 'x')), _p(_opt(gen, 'gen')), _p(alpha), _p(out), B, size,
                                       _stream()), 'mask_blend')
    return alpha, out


def resize_nearest(x, size, out=None, xoff=0):
    x = _dev(x)
    B, C, H, W = x.shape
    Ho, Wo = (size, size) if isinstance(size, int) else size
    if out is None:
        out = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_resize_nearest(_p(x), _p(out), B * C, H, W, Ho, Wo, out.shape[-1], xoff, _stream()), 'resize_nearest')
    return out


def avgpool(x, size):
    """AdaptiveAvgPool2d(size) (restyle arch :89; feature_style_encoder.py:42)."""
    x = _dev(x)
    B, C, H, W = x.shape
    Ho, Wo = (size, size) if isinstance(size, int) else size
    y = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_avgpool(_p(x.contiguous()), _p(y), B * C, H, W, Ho, Wo, _stream()), 'avgpool')
    return y


def resize_bilinear(x, size):
    x = _dev(x)
    B, C, H, W = x.shape
    Ho, Wo = (size, size) if isinstance(size, int) else size
    y = torch.empty(B, C, Ho, Wo, device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_resize_bilinear(_p(x), _p(y), B * C, H, W, Ho, Wo, _stream()), 'resize_bilinear')
    return y


def extract_masks(aligns, size=1024):
    """run_ood_faceGAN_inversion.py:74-87: alpha channel of every level, nearest to 1024, side by side."""
    keys = sorted(aligns)
    B = aligns[keys[0]].shape[0]
    dev = aligns[keys[0]].device
    strip = torch.empty(B, 1, size, size * len(keys), device=dev, dtype=torch.float32)
    for i, k in enumerate(keys):
        a = aligns[k][:, 2:3].contiguous()
        resize_nearest(a, size, out=strip, xoff=size * i)
    return strip


# ----------------------------------------------------------------------------- parameter containers
def BN(depth, bn=True):
    if bn == 'InstanceNorm':
        return nn.InstanceNorm2d(depth, affine=True)
    if bn == 'BatchNorm' or bn is True:
        raise NotImplementedError('BatchNorm bottlenecks belong to the e4e encoder (SURVEY.md §8f N1)')
    return nn.Identity()


class bottleneck_IR(nn.Module):
    """bottleneck_IR(in_channel, depth, stride=1, bn='InstanceNorm', bias=False)."""

    def __init__(self, in_channel, depth, stride=1, bn='InstanceNorm', bias=False):
        super().__init__()
        if stride != 1 or bias or bn not in ('InstanceNorm', False, None):
            raise NotImplementedError('SAMM uses stride-1, bias-free bottlenecks with InstanceNorm (AlignNet) or without a norm (mod_btn)')
        self.in_channel, self.depth = in_channel, depth
        self.norm = bn == 'InstanceNorm'                # bn=False: BN() is nn.Identity (e4e helpers.py:93-99), style_bottleneck_IR
        if in_channel == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(in_channel, depth, (1, 1), stride, bias=False), BN(depth, bn))
        self.res_layer = nn.Sequential(BN(in_channel, bn), nn.Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False),
                                       nn.PReLU(depth), nn.Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BN(depth, bn))
        self._key, self._prep = None, None
        self._chain = {}

    def _prepared(self):
        w1, w2 = self.res_layer[1].weight, self.res_layer[3].weight
        key = (w1.data_ptr(), w1._version, w2.data_ptr(), w2._version)
        if self.in_channel != self.depth:
            w0 = self.shortcut_layer[0].weight
            key = key + (w0.data_ptr(), w0._version)
        if key != self._key:
            if not self.norm:
                self._key, self._prep = key, {'w1': ops.pack_conv3x3(w1.detach()), 'w2': ops.pack_conv3x3(w2.detach())}
                return self._prep
            if self.depth <= 4 and self.in_channel > 8:
                # conv3x3_fewout / conv3x3_small take the raw weights; the round-4 form a transposed copy (+ the 1x1 shortcut's)
                prep = {}
                if self.in_channel % 8 == 0:
                    w11 = self.shortcut_layer[0].weight.detach() if self.in_channel != self.depth else None
                    prep['wt'], prep['w11t'] = fewout_weights(w1.detach(), w11)
                self._key, self._prep = key, prep
                return self._prep
            prep = {'w1': ops.pack_conv3x3(w1.detach())}
            if self.depth > 8:
                prep['w2'] = ops.pack_conv3x3(w2.detach())
            self._key, self._prep = key, prep
        return self._prep

    def _chain_scale(self, B, HW):
        """(mul2 = [2^-e, 2^e], ys_scale (B, depth) = 2^e): power-of-two range scale of PReLU(conv1(InstanceNorm_affine(.))) from a
        bound of the frozen parameters (see forward); one host read per (parameter version, HW), none afterwards."""
        rl = self.res_layer
        ps = (rl[0].weight, rl[0].bias, rl[1].weight, rl[2].weight)
        key = tuple((t.data_ptr(), t._version) for t in ps) + (HW,)
        ent = self._chain.get(key)
        if ent is None:
            g, b_, w1, sl = (t.detach().double() for t in ps)
            xb = (g.abs() * math.sqrt(HW) + b_.abs()).max()
            wb = w1.abs().sum(dim=(1, 2, 3)).max()
            bound = float(xb * wb * sl.abs().max().clamp_min(1.0))
            e = 14 - math.floor(math.log2(bound)) if bound > 0 and math.isfinite(bound) else 0
            e = max(-60, min(60, e))
            mul2 = torch.tensor([2.0 ** -e, 2.0 ** e], device=w1.device, dtype=torch.float32)
            ent = self._chain[key] = {'mul2': mul2, 'e': e, 'ysc': {}}
            if len(self._chain) > 8:
                self._chain.pop(next(iter(self._chain)))
        ysc = ent['ysc'].get(B)
        if ysc is None:
            ysc = ent['ysc'][B] = torch.full((B, self.depth), 2.0 ** ent['e'], device=ent['mul2'].device, dtype=torch.float32)
        return ent['mul2'], ysc

    def forward(self, x, stats=None, want_stats=False):
        """``stats``: InstanceNorm statistics of x when the producer already has them (``want_stats`` of the bottleneck in front):
        the two passes over the 2C-channel tensor that only compute statistics disappear from an AlignNet cycle."""
        prep = self._prepared()
        rl = self.res_layer
        if not self.norm:
            # res_layer = conv3x3 -> PReLU -> conv3x3, plain residual sum (e4e helpers.py:426-452 with BN = Identity)
            r = ops.conv3x3(x, prep['w1'], self.depth, CONV_S1, act=ACT_PRELU, slope=rl[2].weight)
            r = ops.conv3x3(r, prep['w2'], self.depth, CONV_S1)
            shortcut = x if self.in_channel == self.depth else conv1x1(x, self.shortcut_layer[0].weight)
            B = x.shape[0]
            one = torch.ones(B, self.depth, device=r.device, dtype=torch.float32)
            y = affine_apply(r, one, torch.zeros_like(one), res=shortcut)
            return (y, instnorm_stats(y)) if want_stats else y
        sc, sh = instnorm_coeffs(instnorm_stats(x) if stats is None else stats, rl[0].weight, rl[0].bias)
        s1 = None
        if self.in_channel >= 64 and self.depth >= 64:
            # the AlignNet convs (2C -> 2C channels, 7x the generator's FLOPs per image): through the S-form and
            # the 8-wave kernel of the generator's own >= 64-channel layers; InstanceNorm's affine folded into the conversion
            # Range of the two S-form conversions (an f16 pair holds |v| < 65504 and loses its lo half far below 1; the fp32
            # reference has no such limit): the first operand is InstanceNorm's output, |v| <= |gamma|*sqrt(HW) + |beta| by
            # construction; the second is the UN-normalised PReLU(conv) — measured, scaled by a power of two into
            # [512,1024) for the conversion and unscaled in the conv's epilogue (exact).
            B, _, H, W = x.shape
            xs = ops.to_sform(x, sc, shift=sh, out=ops.sform_scratch(B, self.in_channel, H, W, x.device))
            if FUSE_CONV_CHAIN and ops.s1_ys_supported(B, self.in_channel, self.depth, H, W):
                # round 4: the first conv writes the second one's S-form input from its registers — no fp32 r, no range measurement,
                # no conversion pass (three passes over the 2C-channel tensor per bottleneck).  The range scale is a BOUND instead of
                # a measurement: |PReLU(conv(v))| <= max(1,|slope|) * max_m sum|w[m]| * max_c(|gamma_c| sqrt(HW) + |beta_c|), a
                # constant of the frozen weights, scaled to [2^14, 2^15).  The f16 pair then holds every value to 2^-25 of that
                # bound's scaled size, i.e. to ~2^-27 of the tensor's real maximum when the bound is 2^12 above it (measured
                # ratios: 2^9 .. 2^12) — below the fp32 rounding of the 9 * 2C-term sums themselves.
                mul2, ysc = self._chain_scale(B, H * W)
                rs = ops.sform_scratch(B, self.depth, H, W, x.device, tag=3)
                ops.conv3x3(xs, prep['w1'], self.depth, CONV_S1, act=ACT_PRELU, slope=rl[2].weight, ys=rs, ys_scale=ysc, want_y=False)
            else:
                r = ops.conv3x3(xs, prep['w1'], self.depth, CONV_S1, act=ACT_PRELU, slope=rl[2].weight)
                mul2 = ops.absmax_mul2(r)
                rs = ops.to_sform(r, mul2=mul2, out=ops.sform_scratch(B, self.depth, H, W, x.device))
            r = ops.conv3x3(rs, prep['w2'], self.depth, CONV_S1, in_mul2=mul2)
            del xs, rs
        elif self.depth <= 4 and self.in_channel > 8:
            # AlignNet's head: 2C -> 3 -> 3 channels.  Streaming work, exact fp32, K split over the chip
            if 'wt' in prep:
                # one pass over x for the head conv and the 1x1 shortcut conv
                r, s1 = conv3x3_fewout2(x, prep['wt'], self.depth, sc, sh, slope=rl[2].weight, w11t=prep['w11t'], M2=self.depth)
            else:
                r = conv3x3_fewout(x, rl[1].weight, sc, sh, slope=rl[2].weight)
            r = conv3x3_small(r, rl[3].weight)
        else:
            r = ops.conv3x3(x, prep['w1'], self.depth, CONV_S1, in_scale=sc, in_shift=sh, act=ACT_PRELU, slope=rl[2].weight)
            if self.depth > 8:
                r = ops.conv3x3(r, prep['w2'], self.depth, CONV_S1)
            else:
                r = conv3x3_small(r, rl[3].weight)
        if self.in_channel == self.depth:
            shortcut = x
        else:
            s = s1 if s1 is not None else conv1x1(x, self.shortcut_layer[0].weight)
            shortcut = instance_norm(s, self.shortcut_layer[1].weight, self.shortcut_layer[1].bias)
        return instance_norm(r, rl[4].weight, rl[4].bias, res=shortcut, want_stats=want_stats)


def scaleNshiftBlock(in_chn, out_chn, norm_type=False, bias=False):
    return nn.Sequential(bottleneck_IR(in_chn, in_chn, 1, norm_type, bias), bottleneck_IR(in_chn, out_chn, 1, norm_type, bias))


class AlignNet(nn.Module):
    def __init__(self, in_chn, out_chn=3, scale=1., blur_kernel=[1, 3, 3, 1], **kwargs):
        super().__init__()
        self.norm = nn.InstanceNorm2d(in_chn)
        self.body = scaleNshiftBlock(in_chn * 2, out_chn, 'InstanceNorm', kwargs.get('bias', False))
        self.scale = scale
        self.diff_fAndg = kwargs.get('diff_fAndg', True)       # False: the body sees cat([IN(source), IN(target)]) (helpers.py:98-101)

    def forward(self, source, target, st_target=None, **kwargs):
        st_s = instnorm_stats(source)
        st_t = st_target if st_target is not None else instnorm_stats(target)
        if FUSE_STATS:
            # every InstanceNorm's statistics come from the pass that writes its input: the concatenated, normalised pair here, the
            # first bottleneck's output below
            a, st_in = align_input_stats(source, target, st_s, st_t, diff=self.diff_fAndg)
            a, st_a = self.body[0](a, stats=st_in, want_stats=True)
            a = self.body[1](a, stats=st_a)
        else:
            a = align_input(source, target, st_s, st_t, diff=self.diff_fAndg)
            a = self.body[1](self.body[0](a))
        return align_head(a, self.scale)


class SPM_Warp(nn.Module):
    def __init__(self, in_chn, scale=0.1, style_dim=512, blur_kernel=[1, 3, 3, 1], cycle_align=1, **kwargs):
        super().__init__()
        self.body = AlignNet(in_chn, 3, scale=scale, style_dim=style_dim, blur_kernel=blur_kernel, **kwargs)
        for m in self.modules():                        # reference init: helpers.py:117-127
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
        self.scale, self.cycle_align = scale, cycle_align
        self.blur = _BlurBuf(blur_kernel)

    def forward(self, source, target, style=None, aligned=None):
        """source = encoder feature, target = generator feature; returns (aligned_target, field)."""
        cur, acc = target, None
        st_src = instnorm_stats(source)
        for k in range(self.cycle_align):
            a = ops.upfirdn2d(self.body(cur, source, st_target=st_src), self.blur.kernel, pad=(2, 1))
            acc = a if acc is None else field_add(acc, a, self.scale)
            if k == self.cycle_align - 1 and aligned is not None:
                acc = field_upsample_add(aligned, acc)
            cur = warp_blend(target, acc)
        return cur, acc


class _BlurBuf(nn.Module):
    def __init__(self, blur_kernel):
        super().__init__()
        self.register_buffer('kernel', make_kernel(blur_kernel))
        self.pad = (2, 1)


class _NoiseInj(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))


class style_bottleneck_IR(nn.Module):
    """``mod_btn='style_bottleneck_IR'`` (reference src/ops/SAMM/helpers.py:22-40): two norm-free bottlenecks, a ModulatedConv2d on the
    layer's style and a FusedLeakyReLU.  Parameter names follow the reference's state dict."""

    def __init__(self, in_channel, depth, style_dim, stride=1, upsample=False, downsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True, bn=False):
        super().__init__()
        from .modules import FusedLeakyReLU, ModulatedConv2d
        self.btn = nn.Sequential(bottleneck_IR(in_channel, in_channel, stride, bn), bottleneck_IR(in_channel, depth, stride, bn))
        self.final_conv = ModulatedConv2d(depth, depth, 3, style_dim, demodulate=demodulate, upsample=upsample, downsample=downsample,
                                          blur_kernel=blur_kernel)
        self.act = FusedLeakyReLU(depth)

    def forward(self, x, style):
        return self.act(self.final_conv(self.btn(x), style))


class styleBlock(nn.Module):
    """``mod_btn='styleBlock'`` (reference src/ops/SAMM/helpers.py:43-57): two StyledConv, the first without noise, the second with the
    noise / activation flags the caller passes (StyledscaleNshfitBlock: neither)."""

    def __init__(self, in_channel, depth, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1], demodulate=True, noiseInjection=True,
                 activation=False):
        super().__init__()
        from .modules import StyledConv
        self.conv1 = StyledConv(in_channel, depth, 3, style_dim, demodulate=demodulate, upsample=upsample, blur_kernel=blur_kernel,
                                noiseInjection=False, activation=True)
        self.conv2 = StyledConv(depth, depth, 3, style_dim, demodulate=demodulate, upsample=upsample, blur_kernel=blur_kernel,
                                noiseInjection=noiseInjection, activation=activation)

    def forward(self, x, style):
        return self.conv2(self.conv1(x, style), style)


class StyledscaleNshfitBlock(nn.Module):
    """reference src/ops/SAMM/helpers.py:182-216.  btn=None: identity feature extractor (every shipped YAML);
    'style_bottleneck_IR' / 'styleBlock': the modulated extractors (``mod_btn``), after which the alignment works on out_chn channels."""

    def __init__(self, in_chn, out_chn, style_dim, alignment=True, btn=None, **kwargs):
        super().__init__()
        if btn == 'style_bottleneck_IR':
            self.btn1 = style_bottleneck_IR(in_chn, out_chn, style_dim, bn=False)
        elif btn == 'styleBlock':
            self.btn1 = styleBlock(in_chn, out_chn, style_dim, noiseInjection=False, activation=False)
        else:
            self.btn1 = None
            out_chn = in_chn
        if not alignment:
            raise NotImplementedError('alignment=False')
        self.alignment = SPM_Warp(out_chn, **kwargs)
        self.weight = nn.Parameter(torch.ones(1), requires_grad=False)
        self.noiseInj = _NoiseInj()

    def forward(self, x, styles, **kwargs):
        res = x if self.btn1 is None else self.btn1(x, styles)
        transform = kwargs.get('transform', None)           # training-time augmentation hook of the reference (:207-208)
        if transform is not None:
            res = transform(res)
        gen_feat = kwargs.get('image', None)
        assert gen_feat is not None
        return self.alignment(res, gen_feat, styles, kwargs.get('aligned', None))
