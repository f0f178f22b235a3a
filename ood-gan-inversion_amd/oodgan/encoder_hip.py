"""e4e encoder on the HIP kernels — the same parameters, state-dict keys and forward signature as
``oodgan.encoder.Encoder4Editing`` (mirror of reference src/ops/e4e/encoders/psp_encoders.py:125-216), but every
convolution, normalisation, gate application and resize runs through ``liboodgan_hip.so``:

  BatchNorm (eval) before a conv   -> per-channel in_scale / in_shift of ``oodgan_conv3x3`` (shift applied to in-bounds
                                      samples only = BN followed by zero padding)
  BatchNorm (eval) after a conv    -> out_scale + bias of the conv epilogue
  PReLU / LeakyReLU(0.01)          -> per-channel slope in the conv epilogue
  conv3x3 stride 2, pad 1          -> the stride-2 kernel (mode S2) on the input shifted by one zero row / column
  SE gate + residual               -> ``oodgan_instnorm_stats`` (mean), ``oodgan_se_gate``, ``oodgan_affine_apply``
  FPN ``_upsample_add``            -> ``oodgan_resize_bicubic_ac`` (helpers.py:504-521)
  GradualStyleBlock                -> stride-2 convs, all heads side by side (stacked / grouped launches) + ``oodgan_equal_linear``

torch is used for plumbing only (zero-padding copy, strided slice of the shortcut, ReLU / sigmoid on (B,C) vectors).
The encoder runs once per image at 256² and is not part of the measured loop."""
import ctypes
import math

import os

import torch

from . import _lib, ops, samm
from ._lib import ACT_NONE, ACT_PRELU, CONV_S1, CONV_S2, check
from .encoder import Encoder4Editing, ProgressiveBackboneEncoder, fs_encoder_v2


def _bn_affine(bn):
    """eval-mode BatchNorm2d as y = x*sc + sh."""
    sc = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return sc, bn.bias - bn.running_mean * sc


def _rows(v, B):
    return v.detach().float().reshape(1, -1).expand(B, -1).contiguous()


def _pad_tl(x):
    """(B,C,H,W) -> (B,C,H+1,pitch>=W+1) with a zero first row / column; pitch is a multiple of 4 floats."""
    B, C, H, W = x.shape
    pitch = (W + 1 + 3) // 4 * 4
    y = torch.zeros(B, C, H + 1, pitch, device=x.device, dtype=torch.float32)
    y[:, :, 1:, 1:W + 1] = x
    return y, pitch


def _resize_bicubic_ac(x, size, add=None):
    B, C, H, W = x.shape
    y = torch.empty(B, C, size[0], size[1], device=x.device, dtype=torch.float32)
    check(_lib.lib().oodgan_resize_bicubic_ac(ops._p(x.contiguous()), ops._p(None if add is None else add.contiguous()), ops._p(y),
                                              B * C, H, W, size[0], size[1], ops._stream()), 'resize_bicubic_ac')
    return y


class _Packed:
    """packed conv weights, re-packed when the parameter changes"""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, w):
        key = (w.data_ptr(), w._version)
        if key != self.key:
            self.key, self.val = key, ops.pack_conv3x3(w.detach().float().contiguous(), precision='f16s')
        return self.val


TRUNK_TINY_MAX = int(os.environ.get('OODGAN_ENC_TRUNK_TINY_MAX', '32'))     # trunk convs on <= 1024 positions: skinny-GEMM kernel (batch 1-4)
SMALL_S2_SFORM = int(os.environ.get('OODGAN_ENC_SMALL_S2_SFORM', '1'))
HEADS_TINY = int(os.environ.get('OODGAN_HEADS_TINY', '1'))
HEADS_TINY_MAX_OUT = int(os.environ.get('OODGAN_HEADS_TINY_MAX_OUT', '8'))
HEADS_SFORM_MIN_IN = int(os.environ.get('OODGAN_HEADS_SFORM_MIN_IN', '16'))      # grouped head steps with input maps of at least this size go through the S-form (A/B: tools/forward_only.py)


def _conv3x3(x, pk, M, stride=1, **kw):
    if stride == 1:
        return ops.conv3x3(x, pk, M, CONV_S1, **kw)
    assert stride == 2 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
    B, K, H, W = x.shape
    G = kw.get('groups', 1)
    tiny = HEADS_TINY and min(H, W) // 2 <= HEADS_TINY_MAX_OUT and ops.tiny_workspace_bytes(CONV_S2, B, K // G, M, H + 1, W + 1) > 0
    big = ops.s2_fuse_supported(B, K, M, H + 1, W + 1) if G == 1 else \
        (min(H, W) >= HEADS_SFORM_MIN_IN and ops.s2_grouped_supported(B, K // G, M, G, H + 1, W + 1))
    # a trunk conv too small for the 8-wave kernel still takes the S-form route: the 4-wave S-form kernel (what the W+ loop runs on its
    # low-resolution layers) against the fp32-input kernel's exposed global-load latency per K chunk (118 us per conv at B = 1)
    small = SMALL_S2_SFORM and G == 1 and K % 16 == 0 and M >= 64
    if 'in_scale' not in kw and (tiny or big or small):
        # through the phase-split S-form (measured power-of-two range scale; bias + slope in the kernel's epilogue):
        #  * enough work for the 8-wave stride-2 kernel — the stacked first convs of the style heads (512 -> 11 x 512 channels at
        #    64² -> 32²) and the grouped steps that follow while the maps fill a useful part of its 8 x 32 tile.  The fp32-input kernel
        #    ran the stacked conv at 105 TFLOP/s (4.0 ms at batch 8);
        #  * outputs of 8 x 8 and below: the skinny-GEMM kernel (conv_f16s_tiny.hip) — all images' positions packed into the N tiles,
        #    K split over the chip, every weight read once per 128 positions.  The fp32-input kernel spent 0.83 ms on each of the
        #    4² / 2² / 1² steps of the 18 heads at batch 8 (170 MB of weights per step, re-streamed by every image's workgroups).
        mul2 = ops.absmax_mul2(x)
        gp = ops.to_sform_phases(x, H // 2, W // 2, mul2=mul2, pad_tl=True,        # the zero pad on the top / left comes from the conversion
                                 out=ops.sform_phases_scratch(B, K, H // 2, W // 2, x.device))
        return ops.conv3x3(gp, pk, M, CONV_S2, in_mul2=mul2, **kw)
    xp, pitch = _pad_tl(x)
    return ops.conv3x3(xp, pk, M, CONV_S2, in_hw=(H + 1, W + 1), in_pitch=pitch, **kw)


class _HipTrunk:
    """shared HIP implementation of the IR-SE trunk and the GradualStyleBlock heads"""

    def _packed(self, name, w):
        p = self._pk.get(name)
        if p is None:
            p = self._pk[name] = _Packed()
        return p.get(w)

    def _memo(self, key, deps, fn):
        """Derived device constants (folded BatchNorm rows, stacked head weights, ...) are rebuilt only when a parameter
        they come from changes — not with ~600 small torch launches per forward."""
        if not hasattr(self, '_cache'):
            self._cache = {}
        ver = tuple((t.data_ptr(), t._version) for t in deps)
        c = self._cache.get(key)
        if c is None or c[0] != ver:
            c = self._cache[key] = (ver, fn())
        return c[1]

    def _bn_rows(self, key, bn, B):
        """eval-mode BatchNorm2d as per-(b,c) rows (scale, shift) and the (C,) shift."""
        def build():
            sc, sh = _bn_affine(bn)
            return _rows(sc, B), _rows(sh, B), sh.detach().float().contiguous()
        return self._memo((key, B), (bn.weight, bn.bias, bn.running_mean, bn.running_var), build)

    def _const_rows(self, B, C, value, device):
        return self._memo(('const', B, C, value, str(device)), (), lambda: torch.full((B, C), float(value), device=device))

    # ---- one bottleneck_IR(_SE) unit (helpers.py:439-501)
    def _unit(self, idx, u, x):
        B = x.shape[0]
        rl = u.res_layer
        depth = rl[1].weight.shape[0]
        stride = rl[3].stride[0]
        sc1, sh1, _ = self._bn_rows(f'{idx}.bn1', rl[0], B)
        sc2, _, sh2 = self._bn_rows(f'{idx}.bn2', rl[4], B)
        cin = x.shape[1]
        if cin >= 64 and depth >= 64:
            # the stride-1 convs through the S-form (BatchNorm as scale AND shift of the conversion, the border stays zero =
            # norm followed by zero padding) and the LDS-DMA kernels of the generator: at 32² / 16² the fp32-input kernel is 16-32
            # workgroups each walking its K chunks one exposed global-load latency at a time (140 us per conv at any batch size)
            H, W = x.shape[2], x.shape[3]
            xs = ops.to_sform(x, sc1, shift=sh1, out=ops.sform_scratch(B, cin, H, W, x.device, tag=1))
            r = ops.conv3x3(xs, self._packed(f'{idx}.w1', rl[1].weight), depth, CONV_S1, act=ACT_PRELU, slope=rl[2].weight.detach(),
                            tiny_max=TRUNK_TINY_MAX)
        else:
            r = _conv3x3(x, self._packed(f'{idx}.w1', rl[1].weight), depth, 1, in_scale=sc1, in_shift=sh1,
                         act=ACT_PRELU, slope=rl[2].weight.detach())
        if stride == 1 and depth >= 64:
            # un-normalised PReLU(conv) output: measured power-of-two range scale for the f16 pair, undone in the conv (exact)
            mul2 = ops.absmax_mul2(r)
            rs = ops.to_sform(r, mul2=mul2, out=ops.sform_scratch(B, depth, r.shape[2], r.shape[3], x.device, tag=2))
            r = ops.conv3x3(rs, self._packed(f'{idx}.w2', rl[3].weight), depth, CONV_S1, out_scale=sc2, bias=sh2, in_mul2=mul2,
                            tiny_max=TRUNK_TINY_MAX)
        else:
            r = _conv3x3(r, self._packed(f'{idx}.w2', rl[3].weight), depth, stride, out_scale=sc2, bias=sh2)
        if isinstance(u.shortcut_layer, torch.nn.MaxPool2d):
            sc = x if stride == 1 else x[:, :, ::stride, ::stride].contiguous()
        else:
            xs = x if stride == 1 else x[:, :, ::stride, ::stride].contiguous()
            s = samm.conv1x1(xs, u.shortcut_layer[0].weight.detach())
            a, b, _ = self._bn_rows(f'{idx}.bns', u.shortcut_layer[1], B)
            sc = samm.affine_apply(s, a, b)
        if self._mode == 'ir_se':
            se = rl[5]
            g = samm.se_gate(samm.instnorm_stats(r), se.fc1.weight.detach(), se.fc2.weight.detach())      # one launch for fc1-relu-fc2-sigmoid
            return samm.affine_apply(r, g, self._const_rows(B, depth, 0.0, x.device), res=sc)
        return samm.affine_apply(r, self._const_rows(B, depth, 1.0, x.device), self._const_rows(B, depth, 0.0, x.device), res=sc)

    def _style(self, i, feat):
        return self._style_heads([i], {i: feat})[i]

    def _head_level(self, heads, js):
        """Stacked parameters of conv js[n] of head heads[n]: packed (G*512, 512, 3, 3) weight, (G*512,) bias and slopes."""
        convs = [[m for m in self.styles[i].convs if isinstance(m, torch.nn.Conv2d)][j] for i, j in zip(heads, js)]

        def build():
            w = torch.cat([c.weight.detach().float() for c in convs], 0).contiguous()
            bias = torch.cat([c.bias.detach().float() for c in convs], 0).contiguous()
            return ops.pack_conv3x3(w, precision='f16s'), bias, torch.full((w.shape[0],), 0.01, device=w.device)
        return self._memo(('heads', tuple(heads), tuple(js)), [t for c in convs for t in (c.weight, c.bias)], build)

    def _style_heads(self, heads, feat_of):
        """GradualStyleBlock.forward (psp_encoders.py:14-34) of several heads at once.  A head is a chain of stride-2 3x3
        convs (512 -> 512, LeakyReLU(0.01)) down to 1x1 and an EqualLinear; at B = 1 every conv of the chain is a ~10 MB
        weight stream over a handful of output pixels — 98 launches of a latency-bound kernel when run head by head.  Here
        heads that read the same feature map run their first conv as ONE launch over the stacked output channels, and
        from there on all chains advance together as ONE grouped convolution per step (oodgan_conv_args.groups); the
        chains that start on smaller maps join when the running ones have come down to their size (all chains end at
        1x1, so after every step the live maps have one size): 18 heads = 8 conv launches instead of 98.
        ``feat_of[i]`` is head i's input map.  Returns {i: (B, 512)}."""
        out_c = self.styles[heads[0]].out_c
        nconv = {i: sum(isinstance(m, torch.nn.Conv2d) for m in self.styles[i].convs) for i in heads}
        sets = {}
        for i in heads:                             # heads sharing one input map
            sets.setdefault(id(feat_of[i]), []).append(i)
        steps = max(nconv.values())
        live, nxt, x_all = [], {}, None             # live heads in channel order, their next conv index, (B, G*512, h, w)
        for step in range(steps):
            y = None
            if live:
                pk, bias, slope = self._head_level(live, [nxt[i] for i in live])
                y = _conv3x3(x_all, pk, out_c * len(live), 2, bias=bias, act=ACT_PRELU, slope=slope, groups=len(live))
                for i in live:
                    nxt[i] += 1
            for hs in sets.values():
                if nconv[hs[0]] == steps - step:    # this set's chains start now, on their shared map
                    pk, bias, slope = self._head_level(hs, [0] * len(hs))
                    y0 = _conv3x3(feat_of[hs[0]], pk, out_c * len(hs), 2, bias=bias, act=ACT_PRELU, slope=slope)
                    y = y0 if y is None else torch.cat([y, y0], 1)
                    live = live + hs
                    nxt.update({i: 1 for i in hs})
            x_all = y
        B = x_all.shape[0]
        x_all = x_all.reshape(B, len(live), out_c)
        lins = [self.styles[i].linear for i in live]
        if len(live) > 1 and all(l.lr_mul == lins[0].lr_mul and l.weight.shape == lins[0].weight.shape and l.bias is not None for l in lins):
            # the heads' final EqualLinear layers as ONE launch over the stacked weights (was: a slice copy + a launch per head)
            w, b = self._memo(('head_lin', tuple(live)), [t for l in lins for t in (l.weight, l.bias)],
                              lambda: (torch.stack([l.weight.detach().float() for l in lins]).contiguous(),
                                       torch.stack([l.bias.detach().float() for l in lins]).contiguous()))
            y = ops.equal_linear_grouped(x_all.contiguous(), w, b, lr_mul=lins[0].lr_mul)
            self._stacked_deltas = (y, list(live))         # for callers that combine all heads at once (Encoder4EditingHIP.forward)
            return {i: y[:, g] for g, i in enumerate(live)}
        res = {}
        for g, i in enumerate(live):
            blk = self.styles[i]
            res[i] = ops.equal_linear(x_all[:, g].contiguous(), blk.linear.weight.detach(), blk.linear.bias.detach(), lr_mul=blk.linear.lr_mul)
        return res

    def _input(self, x):
        B = x.shape[0]
        il = self.input_layer
        a, _, b = self._bn_rows('in.bn', il[1], B)
        return _conv3x3(x, self._packed('in', il[0].weight), 64, 1, out_scale=a, bias=b, act=ACT_PRELU, slope=il[2].weight.detach())


class Encoder4EditingHIP(_HipTrunk, Encoder4Editing):
    def __init__(self, num_layers, mode='ir', opts=None, bn=True):
        Encoder4Editing.__init__(self, num_layers, mode, opts, bn)
        self._mode = mode
        self._pk = {}

    @torch.no_grad()
    def forward(self, x, **kwargs):
        if not x.is_cuda:
            raise RuntimeError('Encoder4EditingHIP needs a ROCm tensor (no CPU fallback)')
        x = self._input(x.float().contiguous())
        feats = [x]
        c1 = c2 = c3 = None
        for i, layer in enumerate(self.body):
            x = self._unit(i, layer, x)
            if i == 2:
                feats.append(x)
            if i == 6:
                c1 = x
                feats.append(x)
            elif i == 20:
                c2 = x
                feats.append(x)
            elif i == 23:
                c3 = x
                feats.append(x)
        stage = self.progressive_stage.value
        feat_of = {0: c3}
        features, p2 = c3, None
        for i in range(1, min(stage + 1, self.style_count)):
            if i == self.coarse_ind:
                p2 = _resize_bicubic_ac(c3, c2.shape[-2:], add=samm.conv1x1(c2, self.latlayer1.weight.detach(), self.latlayer1.bias.detach()))
                features = p2
            elif i == self.middle_ind:
                features = _resize_bicubic_ac(p2, c1.shape[-2:], add=samm.conv1x1(c1, self.latlayer2.weight.detach(), self.latlayer2.bias.detach()))
            feat_of[i] = features
        self._stacked_deltas = None
        deltas = self._style_heads(sorted(feat_of), feat_of)
        st = self._stacked_deltas
        if st is not None and len(feat_of) == self.style_count and sorted(st[1]) == list(range(self.style_count)):
            # w[:, 0] = delta_0, w[:, i] = delta_0 + delta_i (psp_encoders.py:198-214) from the stacked head outputs: three launches
            # instead of one add per head
            y, live = st
            order = self._memo(('head_order', tuple(live)), (), lambda: torch.tensor([live.index(i) for i in range(self.style_count)], device=y.device))
            w = y.index_select(1, order)
            w[:, 1:] += w[:, :1].clone()
        else:
            w = deltas[0].repeat(self.style_count, 1, 1).permute(1, 0, 2).contiguous()
            for i in sorted(feat_of):
                if i > 0:
                    w[:, i] += deltas[i]
        if kwargs.get('return_feats', False):
            return w, feats
        return w


class ProgressiveBackboneEncoderHIP(_HipTrunk, ProgressiveBackboneEncoder):
    """ReStyle's encoder (restyle_e4e_encoder.py:37-112) on the HIP kernels."""

    def __init__(self, num_layers, mode='ir', n_styles=18, opts=None):
        ProgressiveBackboneEncoder.__init__(self, num_layers, mode, n_styles, opts)
        self._mode = mode
        self._pk = {}

    @torch.no_grad()
    def forward(self, x, **kwargs):
        if not x.is_cuda:
            raise RuntimeError('ProgressiveBackboneEncoderHIP needs a ROCm tensor (no CPU fallback)')
        x = self._input(x.float().contiguous())
        feats = [x]
        for i, layer in enumerate(self.body):
            x = self._unit(i, layer, x)
            if i in (2, 6, 20, 23):
                feats.append(x)
        heads = list(range(0, min(self.progressive_stage.value + 1, self.style_count)))
        deltas = self._style_heads(heads, {i: x for i in heads})
        w = deltas[0].repeat(self.style_count, 1, 1).permute(1, 0, 2).contiguous()
        for i in heads[1:]:
            w[:, i] += deltas[i]
        if kwargs.get('return_feats', False):
            return w, feats
        return w


class fs_encoder_v2HIP(_HipTrunk, fs_encoder_v2):
    """Feature-Style encoder (feature_style_encoder.py:12-74) on the HIP kernels: every IBasicBlock is two ``oodgan_conv3x3``
    launches with the three BatchNorms and the PReLU folded in, the shortcut an ``oodgan_conv1x1`` + affine, the pooled
    descriptors ``oodgan_avgpool`` and the ``n_styles`` heads ONE ``oodgan_equal_linear`` (scale 1) over the stacked
    (n_styles*512, 8640) weight."""

    def __init__(self, n_styles=18, opts=None, stride=(1, 1), **kwargs):
        fs_encoder_v2.__init__(self, n_styles, opts, stride, **kwargs)
        self._pk = {}
        self._heads = None

    def _block(self, name, blk, x):
        B = x.shape[0]
        cout = blk.conv1.weight.shape[0]
        a1, b1 = _bn_affine(blk.bn1)
        a2, b2 = _bn_affine(blk.bn2)
        r = _conv3x3(x, self._packed(name + '.c1', blk.conv1.weight), cout, 1, in_scale=_rows(a1, B), in_shift=_rows(b1, B),
                     out_scale=_rows(a2, B), bias=b2.detach().contiguous(), act=ACT_PRELU, slope=blk.prelu.weight.detach())
        a3, b3 = _bn_affine(blk.bn3)
        r = _conv3x3(r, self._packed(name + '.c2', blk.conv2.weight), cout, blk.stride, out_scale=_rows(a3, B), bias=b3.detach().contiguous())
        if blk.downsample is None:
            idt = x
        else:
            xs = x if blk.stride == 1 else x[:, :, ::blk.stride, ::blk.stride].contiguous()
            a, b = _bn_affine(blk.downsample[1])
            idt = samm.affine_apply(samm.conv1x1(xs, blk.downsample[0].weight.detach()), _rows(a, B), _rows(b, B))
        return samm.affine_apply(r, torch.ones(B, cout, device=x.device), torch.zeros(B, cout, device=x.device), res=idt)

    def _stage(self, name, stage, x):
        for i, blk in enumerate(stage):
            x = self._block(f'{name}.{i}', blk, x)
        return x

    def _content(self, x):
        B = x.shape[0]
        cl = self.content_layer
        a0, b0 = _bn_affine(cl[0])
        a2, b2 = _bn_affine(cl[2])
        r = _conv3x3(x, self._packed('cl.1', cl[1].weight), 512, 1, in_scale=_rows(a0, B), in_shift=_rows(b0, B), out_scale=_rows(a2, B),
                     bias=b2.detach().contiguous(), act=ACT_PRELU, slope=cl[3].weight.detach())
        a5, b5 = _bn_affine(cl[5])
        return _conv3x3(r, self._packed('cl.4', cl[4].weight), 512, cl[4].stride[0], out_scale=_rows(a5, B), bias=b5.detach().contiguous())

    def _head_weights(self):
        key = tuple((s.weight.data_ptr(), s.weight._version, s.bias._version) for s in self.styles)
        if self._heads is None or self._heads[0] != key:
            self._heads = (key, torch.cat([s.weight.detach() for s in self.styles], 0).contiguous(),
                           torch.cat([s.bias.detach() for s in self.styles], 0).contiguous())
        return self._heads[1], self._heads[2]

    @torch.no_grad()
    def forward(self, x, return_feats=False):
        if not x.is_cuda:
            raise RuntimeError('fs_encoder_v2HIP needs a ROCm tensor (no CPU fallback)')
        x = x.float().contiguous()
        B = x.shape[0]
        a, b = _bn_affine(self.conv[1])
        x = _conv3x3(x, self._packed('stem', self.conv[0].weight), 64, 1, out_scale=_rows(a, B), bias=b.detach().contiguous(), act=ACT_PRELU,
                     slope=self.conv[2].weight.detach())
        taps, pooled = [x], []
        x = self._stage('b1', self.block_1, x)
        taps.append(x)
        pooled.append(samm.avgpool(x, 3))
        x = self._stage('b2', self.block_2, x)
        taps.append(x)
        pooled.append(samm.avgpool(x, 3))
        x = self._stage('b3', self.block_3, x)
        taps.append(x)
        content = self._content(x)
        pooled.append(samm.avgpool(x, 3))
        x = self._stage('b4', self.block_4, x)
        pooled.append(samm.avgpool(x, 3))
        d = torch.cat(pooled, dim=1).flatten(1).contiguous()
        w, bias = self._head_weights()
        out = ops.linear(d, w, bias).reshape(B, len(self.styles), -1)
        return (out, content, taps) if return_feats else (out, content)
