"""The skinny-GEMM kernel of the 4x4 / 8x8 layers (csrc/conv_f16s_tiny.hip: images packed into the N tiles, K split over workgroups,
fixed-order combine by the last workgroup of a tile) against the tile kernels it replaces — which are pinned against the oracle
and the reference's golden vectors (test_hip_ops.py) — at the geometries of the W+ loop: stride 1 forward (noise + bias +
lrelu) and input gradient (style-gradient dot), stride 2 on the phase-split S-form (input gradient of the up-conv)."""
import math
import os
import sys

import pytest
import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _both(fn):
    from oodgan import ops
    old = ops.USE_TINY
    try:
        ops.USE_TINY = False
        ref = fn()
        ops.USE_TINY = True
        new = fn()
    finally:
        ops.USE_TINY = old
    return ref, new


@pytest.mark.parametrize('B,C,M,H', [(8, 512, 512, 4), (8, 512, 512, 8), (3, 128, 64, 8), (1, 64, 96, 4), (5, 512, 512, 4)])
def test_stride1_forward_and_input_gradient(B, C, M, H):
    from oodgan import _lib, ops
    dev = torch.device('cuda:0')
    assert _lib.lib().oodgan_conv3x3_tiny_workspace(ops.CONV_S1, B, C, M, H, H) > 0
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    d = (1 + 0.3 * torch.randn(B, M, generator=g)).abs().to(dev)
    w = (torch.randn(M, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
    wf = ops.pack_conv3x3(w, precision='f16s')
    nz = torch.randn(B, 1, H, H, generator=g).to(dev)
    nw, bias = torch.tensor([0.3], device=dev), (0.1 * torch.randn(M, generator=g)).to(dev)
    xs = ops.to_sform(x, s)
    y0, y1 = _both(lambda: ops.conv3x3(xs, wf, M, ops.CONV_S1, out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU))
    assert _rel(y1, y0) < 5e-6
    y0, y1 = _both(lambda: ops.conv3x3(xs, wf, M, ops.CONV_S1, out_scale=d, noise=nz[:1], noise_weight=nw, act=ops.ACT_LRELU))
    assert _rel(y1, y0) < 5e-6
    # input gradient: transposed weights, dot with the saved forward input, range scale
    if C == M:
        wb = ops.pack_conv3x3(w, transpose=True, flip=True, precision='f16s')
        gsrc = (1e-3 * torch.randn(B, M, H, H, generator=g)).to(dev)
        mul2 = torch.tensor([2.0 ** -13, 2.0 ** 13], device=dev)
        gs = ops.to_sform(gsrc, d, mul2)
        (dx0, dot0), (dx1, dot1) = _both(lambda: ops.conv3x3(gs, wb, C, ops.CONV_S1, out_scale=s, dotx=x, in_mul2=mul2))
        assert _rel(dx1, dx0) < 5e-6 and _rel(dot1, dot0) < 1e-5       # another order of the 4608-term fp32 sums (K split 8-18 ways)
    # a second launch on the same workspace (the counters were left at zero)
    y2 = ops.conv3x3(xs, wf, M, ops.CONV_S1, out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU)
    y3 = ops.conv3x3(xs, wf, M, ops.CONV_S1, out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU)
    assert torch.equal(y2, y3)


@pytest.mark.parametrize('B,Cg,Cx,H', [(8, 512, 512, 4), (8, 512, 512, 8), (2, 64, 128, 8), (3, 128, 64, 4)])
def test_stride2_input_gradient_of_the_up_conv(B, Cg, Cx, H):
    """g2 (B, Cg, 2H+1, 2H+1) in phase-split S-form -> dx (B, Cx, H, H) with the style-gradient dot."""
    from oodgan import _lib, ops
    dev = torch.device('cuda:0')
    Hin = 2 * H + 1
    assert _lib.lib().oodgan_conv3x3_tiny_workspace(ops.CONV_S2, B, Cg, Cx, Hin, Hin) > 0
    g = torch.Generator().manual_seed(B * 10 + H)
    P2 = (Hin + 3) // 4 * 4
    g2 = (1e-3 * torch.randn(B, Cg, Hin, P2, generator=g)).to(dev)
    d = (1 + 0.3 * torch.randn(B, Cg, generator=g)).to(dev)
    s = (1 + 0.3 * torch.randn(B, Cx, generator=g)).to(dev)
    x = torch.randn(B, Cx, H, H, generator=g).to(dev)
    w = (torch.randn(Cg, Cx, 3, 3, generator=g) / math.sqrt(Cx * 9)).to(dev)
    wb = ops.pack_conv3x3(w, transpose=True, flip=False, precision='f16s')
    mul2 = torch.tensor([2.0 ** -13, 2.0 ** 13], device=dev)
    gp = ops.to_sform_phases(g2, H, H, d, mul2, in_pitch=P2)
    (dx0, dot0), (dx1, dot1) = _both(lambda: ops.conv3x3(gp, wb, Cx, ops.CONV_S2, out_scale=s, dotx=x, in_mul2=mul2))
    assert _rel(dx1, dx0) < 5e-6 and _rel(dot1, dot0) < 1e-5       # another order of the 4608-term fp32 sums (K split 8-18 ways)


@pytest.mark.parametrize('B,G,K,Mg,Ho', [(8, 3, 64, 64, 4), (8, 18, 128, 32, 1), (1, 4, 64, 96, 2), (3, 5, 96, 64, 8), (2, 2, 512, 512, 1),
                                         (5, 1, 64, 64, 2)])
def test_stride2_grouped_forward_down_to_1x1(B, G, K, Mg, Ho):
    """Round 4: the style heads of the e4e encoder (GradualStyleBlock, psp_encoders.py:14-34 — chains of Conv2d(512, 512, 3, 2, 1) +
    LeakyReLU(0.01) down to 1x1, all heads as one grouped conv per step) on the skinny-GEMM kernel: groups, PReLU slopes and bias in the
    finishing pass, 2x2 and 1x1 outputs — against nn.functional.conv2d(groups=G) and against the fp32-input grouped kernel."""
    import torch.nn.functional as F
    from oodgan import _lib, ops
    dev = torch.device('cuda:0')
    H = 2 * Ho
    g = torch.Generator().manual_seed(B * 1000 + G * 10 + Ho)
    x = (20.0 * torch.randn(B, G * K, H, H, generator=g))
    w = torch.randn(G * Mg, K, 3, 3, generator=g) / math.sqrt(K * 9)
    bias = torch.randn(G * Mg, generator=g)
    slope = 0.05 + 0.1 * torch.rand(G * Mg, generator=g)
    ref = F.prelu(F.conv2d(x, w, bias, stride=2, padding=1, groups=G), slope)
    pitch = (H + 1 + 3) // 4 * 4
    xp = torch.zeros(B, G * K, H + 1, pitch)
    xp[:, :, 1:, 1:H + 1] = x
    xp = xp.to(dev)
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    assert ops.tiny_workspace_bytes(ops.CONV_S2, B, K, G * Mg, H + 1, H + 1) > 0
    mul2 = ops.absmax_mul2(x.to(dev))
    gp = ops.to_sform_phases(xp, Ho, Ho, mul2=mul2, in_pitch=pitch)
    kw = dict(bias=bias.to(dev), act=ops.ACT_PRELU, slope=slope.to(dev), groups=G)
    _lib.dispatch_reset()
    y = ops.conv3x3(gp, wpk, G * Mg, ops.CONV_S2, in_mul2=mul2, **kw)
    assert _lib.dispatch_count('tiny') == 1
    assert y.shape == ref.shape
    assert _rel(y.cpu(), ref) < 2e-5
    if Mg % 64 == 0:
        y2 = ops.conv3x3(xp, wpk, G * Mg, ops.CONV_S2, in_hw=(H + 1, H + 1), in_pitch=pitch, **kw)      # fp32-input grouped kernel
        assert _rel(y, y2) < 2e-5
    assert torch.equal(ops.conv3x3(gp, wpk, G * Mg, ops.CONV_S2, in_mul2=mul2, **kw), y)


@pytest.mark.parametrize('B,C,M,H', [(1, 256, 256, 32), (1, 512, 512, 16), (4, 128, 64, 16), (2, 64, 96, 16)])
def test_stride1_mid_size_maps_of_the_encoder_trunk(B, C, M, H):
    """Round 4: 16x16 / 32x32 maps with at most 1024 positions (the IR-SE50 trunk of the e4e encoder at batch 1-4, helpers.py:479-501)
    on the skinny-GEMM kernel — opt-in through ``tiny_max`` — with PReLU slopes or the folded BatchNorm (out_scale, bias) in the finishing pass."""
    import torch.nn.functional as F
    from oodgan import _lib, ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B * 100 + H + C)
    x = torch.randn(B, C, H, H, generator=g)
    sc, sh = 1 + 0.3 * torch.randn(B, C, generator=g), 0.2 * torch.randn(B, C, generator=g)
    w = torch.randn(M, C, 3, 3, generator=g) / math.sqrt(C * 9)
    slope = 0.05 + 0.2 * torch.rand(M, generator=g)
    d, bias = 1 + 0.3 * torch.randn(B, M, generator=g), torch.randn(M, generator=g)
    xn = x * sc[:, :, None, None] + sh[:, :, None, None]
    xs = ops.to_sform(x.to(dev), sc.to(dev), shift=sh.to(dev))
    wf = ops.pack_conv3x3(w.to(dev), precision='f16s')
    _lib.dispatch_reset()
    y = ops.conv3x3(xs, wf, M, ops.CONV_S1, act=ops.ACT_PRELU, slope=slope.to(dev), tiny_max=32)
    z = ops.conv3x3(xs, wf, M, ops.CONV_S1, out_scale=d.to(dev), bias=bias.to(dev), tiny_max=32)
    assert _lib.dispatch_count('tiny') == 2
    raw = F.conv2d(xn, w, padding=1)
    assert _rel(y.cpu(), F.prelu(raw, slope)) < 2e-5
    assert _rel(z.cpu(), raw * d[:, :, None, None] + bias[None, :, None, None]) < 2e-5
    y0 = ops.conv3x3(xs, wf, M, ops.CONV_S1, act=ops.ACT_PRELU, slope=slope.to(dev))            # the tile kernels (default tiny_max)
    assert _lib.dispatch_count('tiny') == 2 and _rel(y, y0) < 5e-6
