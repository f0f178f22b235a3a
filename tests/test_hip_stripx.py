"""The strip conv with in-kernel conversion of an F-form input (csrc/conv_f16s_stripx.hip, oodgan_conv_args.x_fform) against the
two-pass path it replaces at the 1024² level of the W+ loop: S-form producer + strip conv (conv_f16s_strip.hip), both already
pinned against the oracle / the reference's float64 generator (test_hip_ops.py, test_hip_wplus_golden.py).  Same arithmetic up to
the order of the fp32 accumulation of the hi*lo / lo*hi terms."""
import math
import os
import sys

import pytest
import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))

pytestmark = pytest.mark.gpu


def _to_fform(x):
    B, C, H, W = x.shape
    return x.view(B, C // 16, 16, H, W).permute(0, 1, 3, 4, 2).contiguous().view(B, C, H, W)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('waves', [8, 4])
@pytest.mark.parametrize('B,H,W', [(2, 64, 64), (1, 40, 96), (3, 8, 32), (1, 1024, 1024), (8, 256, 32)])
def test_forward_with_fform_input(B, H, W, waves, tunable):
    """waves = 4: the one-wave-per-SIMD kernel of round 3 (the default); waves = 8: csrc/experimental/conv_f16s_stripx8.hip (two waves per SIMD:
    K split over a wave pair, producer / finisher roles) — measured equal in round 5 and since round 6 only in a library built with
    `make STRIPX8=1`: the default library must refuse the tunable's value loudly.  Both against the S-form strip kernel."""
    from oodgan import ops
    tunable('stripx_waves', waves)
    if waves == 8:
        with open(os.path.join(R_, 'ood-gan-inversion_amd', 'oodgan', 'liboodgan_hip.so'), 'rb') as f:
            has8 = b'conv_f16s_stripx8_fwd_kernel' in f.read()          # the kernel's symbol: only in a `make STRIPX8=1` library
        if not has8:
            dev = torch.device('cuda:0')
            xf = ops.FForm(torch.zeros(1, 32, 8, 32, device=dev))
            wf = ops.pack_conv3x3(torch.zeros(32, 32, 3, 3, device=dev), precision='f16s')
            with pytest.raises(RuntimeError, match='STRIPX8'):
                ops.conv3x3(xf, wf, 32, ops.CONV_S1, in_scale=torch.ones(1, 32, device=dev), out_scale=torch.ones(1, 32, device=dev))
            return
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(H * 7 + W)
    C = 32
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    d = (1 + 0.3 * torch.randn(B, C, generator=g)).abs().to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
    wf = ops.pack_conv3x3(w, precision='f16s')
    nz = torch.randn(B, 1, H, W, generator=g).to(dev)
    nw, bias = torch.tensor([0.3], device=dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    w_rgb, s_rgb = torch.randn(3, C, generator=g).to(dev), (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    assert ops.xf_supported(B, C, C, H, W)
    xs = ops.to_sform(x, s)
    y0, rgb0 = ops.conv3x3(xs, wf, C, ops.CONV_S1, out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU, rgb=(w_rgb, s_rgb))
    y1, rgb1 = ops.conv3x3(ops.FForm(_to_fform(x)), wf, C, ops.CONV_S1, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_weight=nw,
                           act=ops.ACT_LRELU, rgb=(w_rgb, s_rgb))
    assert isinstance(y1, ops.FForm)
    assert _rel(y1.to_nchw(), y0) < 1e-6 and _rel(rgb1, rgb0) < 1e-6
    y2 = ops.conv3x3(ops.FForm(_to_fform(x)), wf, C, ops.CONV_S1, in_scale=s, out_scale=d, bias=bias, noise=nz[:1], noise_weight=nw, act=ops.ACT_LRELU)
    y3 = ops.conv3x3(xs, wf, C, ops.CONV_S1, out_scale=d, bias=bias, noise=nz[:1], noise_weight=nw, act=ops.ACT_LRELU)
    assert _rel(y2.to_nchw(), y3) < 1e-6


@pytest.mark.parametrize('B,H,W', [(2, 64, 64), (1, 40, 96), (3, 8, 32), (1, 1024, 1024)])
@pytest.mark.parametrize('pre', [True, False])
def test_input_gradient_with_the_activation_backward_inside(B, H, W, pre):
    from oodgan import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(H * 5 + W + 1)
    C = 32
    out2 = torch.randn(B, C, H, W, generator=g).to(dev)
    out1 = torch.randn(B, C, H, W, generator=g).to(dev)
    g_rgb = (1e-3 * torch.randn(B, 3, H, W, generator=g)).to(dev)
    nz = torch.randn(B, 1, H, W, generator=g).to(dev)
    nw, bias = torch.tensor([0.3], device=dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    d = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    s1 = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    w_rgb, s_rgb = torch.randn(3, C, generator=g).to(dev), (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
    wb = ops.pack_conv3x3(w, transpose=True, flip=True, precision='f16s')
    mul2 = torch.tensor([2.0 ** -14, 2.0 ** 14], device=dev)
    o2f, o1f = ops.FForm(_to_fform(out2)), ops.FForm(_to_fform(out1))
    # two passes: producer -> S-form -> strip conv
    gin = ops.SForm(B, C, H, W, dev)
    r0, t0, pm0 = ops.act_bwd_producer(o2f, None, nz, nw, bias, d, mul2, gin, g_rgb=g_rgb, w_rgb=w_rgb, s_rgb=s_rgb)
    dx0, dot0 = ops.conv3x3(gin, wb, C, ops.CONV_S1, out_scale=s1, dotx=out1, in_mul2=mul2, dot_actgrad=ops.DotActGrad() if pre else None)
    # one pass
    xa = ops.ActBwdX(nz, nw, bias, d, mul2, g_rgb, w_rgb, s_rgb)
    dx1, dot1 = ops.conv3x3(o2f, wb, C, ops.CONV_S1, out_scale=s1, dotx=o1f, in_mul2=mul2, dot_actgrad=ops.DotActGrad() if pre else None, xf_act=xa)
    assert _rel(dx1, dx0) < 1e-6
    assert _rel(dot1, dot0) < 2e-5          # sums of +-1e-3 terms over H*W pixels in a different order
    assert _rel(xa.r, r0) < 2e-5 and _rel(xa.t, t0) < 2e-5
    # the recorded maximum feeds a power-of-two range scale: s_rgb is folded into the slope here, one rounding apart from the producer
    assert abs(float(xa.part_m.max()) - float(pm0.max())) <= 1e-6 * float(pm0.max())
    # no noise / shared noise variants
    xa2 = ops.ActBwdX(None, None, None, d, mul2, g_rgb, w_rgb, s_rgb)
    dx2, _ = ops.conv3x3(o2f, wb, C, ops.CONV_S1, out_scale=s1, dotx=o1f, in_mul2=mul2, xf_act=xa2)
    gin2 = ops.SForm(B, C, H, W, dev)
    r2, _, _ = ops.act_bwd_producer(o2f, None, None, None, None, d, mul2, gin2, g_rgb=g_rgb, w_rgb=w_rgb, s_rgb=s_rgb)
    dx3, _ = ops.conv3x3(gin2, wb, C, ops.CONV_S1, out_scale=s1, dotx=out1, in_mul2=mul2)
    assert _rel(dx2, dx3) < 1e-6 and _rel(xa2.r, r2) < 2e-5


@pytest.mark.parametrize('B,C,H,W', [(2, 32, 48, 48), (1, 32, 32, 80), (1, 16, 64, 32), (1, 32, 512, 512)])
@pytest.mark.parametrize('rank_one', [False, True])
def test_blur_act_fform_equals_blur_act_sform(B, C, H, W, rank_one):
    """F-form tail of the up-conv (tile kernel, and the strip walk selected by rank_one) against the S-form / NCHW producer."""
    from oodgan import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(77 + H + W)
    pitch = (2 * W + 1 + 3) // 4 * 4
    z = torch.randn(B, C, 2 * H + 1, pitch, generator=g).to(dev)
    z[..., 2 * W + 1:] = float('nan')                     # columns between the valid width and the pitch are not defined
    k1 = torch.tensor([1., 3., 3., 1.])
    k = (k1[:, None] * k1[None, :] / 64 * 4).contiguous().to(dev)
    nz = torch.randn(B, 1, 2 * H, 2 * W, generator=g).to(dev)
    nw, bias = torch.tensor([0.3], device=dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    ys = ops.SForm(B, C, 2 * H, 2 * W, dev)
    vm0 = torch.zeros(B, ops.VMAX_SLOTS, dtype=torch.int32, device=dev)
    vm1 = torch.zeros_like(vm0)
    y0 = ops.blur_act_sform(z, k, H, W, bias, nz, nw, act=True, ys=ys, ys_scale=s, vmax=vm0)
    y1 = ops.blur_act_fform(z, k, H, W, bias, nz, nw, act=True, ys_scale=s, vmax=vm1, rank_one=rank_one)
    if rank_one:        # separable evaluation: another order of the 16 products
        assert _rel(y1.to_nchw(), y0) < 2e-6
        a, b_ = vm0.max(dim=1).values.view(torch.float32), vm1.max(dim=1).values.view(torch.float32)
        assert float(((a - b_).abs() / a).max()) < 2e-6
    else:
        assert torch.equal(y1.to_nchw(), y0)
        assert torch.equal(vm0.max(dim=1).values, vm1.max(dim=1).values)
    # shared noise, no bias
    y2 = ops.blur_act_sform(z, k, H, W, None, nz[:1], nw, act=True, ys=ys, ys_scale=s)
    y3 = ops.blur_act_fform(z, k, H, W, None, nz[:1], nw, act=True, ys_scale=s, rank_one=rank_one)
    assert _rel(y3.to_nchw(), y2) < 2e-6


@pytest.mark.parametrize('B,C,H,W', [(2, 32, 48, 48), (1, 32, 32, 80), (1, 16, 64, 32), (1, 64, 40, 100), (1, 32, 512, 512)])
def test_blur_act_sform_strip_walk_equals_tile_kernel(B, C, H, W):
    """S-form + NCHW tail of the up-conv: the strip walk (rank_one promise) against the tile kernel; its S-form output is exactly
    to_sform(its own y, style); partial last strips (2W % 64 != 0), shared noise, no bias."""
    from oodgan import ops
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(177 + H + W)
    pitch = (2 * W + 1 + 3) // 4 * 4
    z = torch.randn(B, C, 2 * H + 1, pitch, generator=g).to(dev)
    z[..., 2 * W + 1:] = float('nan')                     # columns between the valid width and the pitch are not defined
    k1 = torch.tensor([1., 3., 3., 1.])
    k = (k1[:, None] * k1[None, :] / 64 * 4).contiguous().to(dev)
    nz = torch.randn(B, 1, 2 * H, 2 * W, generator=g).to(dev)
    nw, bias = torch.tensor([0.3], device=dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    ys0, ys1 = ops.SForm(B, C, 2 * H, 2 * W, dev), ops.SForm(B, C, 2 * H, 2 * W, dev)
    vm0 = torch.zeros(B, ops.VMAX_SLOTS, dtype=torch.int32, device=dev)
    vm1 = torch.zeros_like(vm0)
    y0 = ops.blur_act_sform(z, k, H, W, bias, nz, nw, act=True, ys=ys0, ys_scale=s, vmax=vm0)
    y1 = ops.blur_act_sform(z, k, H, W, bias, nz, nw, act=True, ys=ys1, ys_scale=s, vmax=vm1, rank_one=True)
    assert _rel(y1, y0) < 2e-6                            # separable evaluation in another order
    assert torch.equal(ys1.data, ops.to_sform(y1, s).data)
    want = (y1 * s[:, :, None, None]).abs().amax(dim=(1, 2, 3))
    assert torch.equal(vm1.view(torch.float32).amax(dim=1), want)
    y2 = ops.blur_act_sform(z, k, H, W, None, nz[:1], nw, act=False, ys=ys0, ys_scale=s)
    y3 = ops.blur_act_sform(z, k, H, W, None, nz[:1], nw, act=False, ys=ys1, ys_scale=s, rank_one=True)
    assert _rel(y3, y2) < 2e-6
    assert torch.equal(ys1.data, ops.to_sform(y3, s).data)


@pytest.mark.parametrize('taps', [[0., 1., 1., 0.], [0., 2., 1., 0.5], [1., 0., 0., 3.]])
def test_blur_strip_walk_with_a_zero_corner_kernel(taps):
    """A rank-one blur kernel whose corner tap is zero (the strip walk divided the column taps by kern[15] and
    produced NaN): the strip walk picks its largest tap as pivot and still equals the tile kernel."""
    from oodgan import ops
    dev = torch.device('cuda:0')
    B, C, H, W = 1, 32, 48, 64
    g = torch.Generator().manual_seed(5)
    pitch = (2 * W + 1 + 3) // 4 * 4
    z = torch.randn(B, C, 2 * H + 1, pitch, generator=g).to(dev)
    k1 = torch.tensor(taps)
    k2 = torch.tensor(taps[::-1]) * 0.5 if taps[0] else k1
    k = (k1[:, None] * k2[None, :]).contiguous().to(dev)
    nz = torch.randn(B, 1, 2 * H, 2 * W, generator=g).to(dev)
    nw, bias = torch.tensor([0.3], device=dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    ys0, ys1 = ops.SForm(B, C, 2 * H, 2 * W, dev), ops.SForm(B, C, 2 * H, 2 * W, dev)
    y0 = ops.blur_act_sform(z, k, H, W, bias, nz, nw, act=True, ys=ys0, ys_scale=s)
    y1 = ops.blur_act_sform(z, k, H, W, bias, nz, nw, act=True, ys=ys1, ys_scale=s, rank_one=True)
    y2 = ops.blur_act_fform(z, k, H, W, bias, nz, nw, act=True, ys_scale=s, rank_one=True)
    assert torch.isfinite(y1).all() and torch.isfinite(y2.to_nchw()).all()
    assert _rel(y1, y0) < 2e-6 and _rel(y2.to_nchw(), y0) < 2e-6


@pytest.mark.parametrize('waves', [12, 6, 4])
@pytest.mark.parametrize('B,K,M,H,W', [(2, 64, 32, 32, 32), (1, 32, 32, 20, 45), (1, 16, 64, 7, 30), (2, 64, 32, 64, 100), (1, 64, 32, 512, 512)])
def test_upconv_vblur_one_pass_equals_transposed_conv_then_blur(B, K, M, H, W, waves, tunable):
    """conv_f16s_upvb.hip (transposed conv + Blur + noise + bias + lrelu in one pass, the blur's vertical pass folded into the weights)
    against the two-pass path it replaces — conv3x3(mode T2) then blur_act_fform — on the same S-form input: image, recorded range
    maximum, shared noise / no bias / no activation; ragged sizes (partial last tiles in both directions), two channel blocks."""
    from oodgan import ops, _lib
    tunable('upvb_waves', waves)        # the persistent 12-wave form and the two tile forms
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(500 + H + W + K)
    x = torch.randn(B, K, H, W, generator=g).to(dev)
    s = (1 + 0.3 * torch.randn(B, K, generator=g)).to(dev)
    d = (1 + 0.3 * torch.randn(B, M, generator=g)).abs().to(dev)
    w = (torch.randn(M, K, 3, 3, generator=g)).to(dev)
    scale = 1.0 / math.sqrt(K * 9)
    k1 = torch.tensor([1., 3., 3., 1.])
    k = (k1[:, None] * k1[None, :] / 64 * 4).contiguous().to(dev)
    nz = torch.randn(B, 1, 2 * H, 2 * W, generator=g).to(dev)
    nw, bias = torch.tensor([0.3], device=dev), (0.1 * torch.randn(M, generator=g)).to(dev)
    ysc = (1 + 0.3 * torch.randn(B, M, generator=g)).to(dev)
    xs = ops.to_sform(x, s)
    wpk = ops.pack_conv3x3(w, scale, transpose=False, flip=False, precision='f16s')
    wvb = ops.pack_upconv_vblur(w, scale, k)
    assert wvb is not None and ops.upconv_vblur_supported(B, K, M, H, W)
    for (b_, n_, act) in ((bias, nz, True), (None, nz[:1], False), (bias, None, True)):
        vm0 = torch.zeros(B, ops.VMAX_SLOTS, dtype=torch.int32, device=dev)
        vm1 = torch.zeros_like(vm0)
        z = ops.conv3x3(xs, wpk, M, ops.CONV_T2, out_scale=d)
        y0 = ops.blur_act_fform(z, k, H, W, b_, n_, nw, act=act, ys_scale=ysc, vmax=vm0, rank_one=True).to_nchw()
        _lib.dispatch_reset()
        y1 = ops.upconv_vblur_fform(xs, wvb, out_scale=d, bias=b_, noise=n_, noise_weight=nw, act=act, ys_scale=ysc, vmax=vm1)
        assert _lib.dispatch_count('upvb') == 1
        y1 = y1.to_nchw()
        assert torch.isfinite(y1).all()
        err = (y1 - y0).abs().max().item() / y0.abs().max().item()
        assert err < 3e-6, (err, act)
        a, b2 = vm0.max(dim=1).values.view(torch.float32), vm1.max(dim=1).values.view(torch.float32)
        assert float(((a - b2).abs() / a).max()) < 3e-6
    # a kernel that is not an outer product has no one-pass form
    assert ops.pack_upconv_vblur(w, scale, torch.eye(4, device=dev)) is None


@pytest.mark.parametrize('waves', [12, 4])
@pytest.mark.parametrize('B,K,M,H,W', [(2, 64, 32, 20, 45), (1, 16, 32, 7, 30), (1, 64, 32, 128, 128)])
def test_upconv_vblur_one_pass_vs_oracle(B, K, M, H, W, waves, tunable):
    """conv_f16s_upvb.hip DIRECTLY against the oracle's up-sampling StyledConv (oracle/ref_cpu.py: modulated_conv2d(upsample=True) —
    conv_transpose2d + Blur, reference model.py:247-258,199-205 — then NoiseInjection + FusedLeakyReLU, model.py:343-350,283-292) on
    ragged shapes: style, demodulation, folded vertical blur, horizontal pass on the accumulators, noise, bias, activation — nothing
    of the HIP two-pass path in between."""
    from oodgan import ops, synth, _lib
    from oracle import ref_cpu as R
    tunable('upvb_waves', waves)
    dev = torch.device('cuda:0')
    S = 64
    P = {}
    synth._styled_conv(P, 'q', K, M, S, 31 + H, True, 0.1)
    g = torch.Generator().manual_seed(900 + H + W + K)
    x = torch.randn(B, K, H, W, generator=g)
    wlat = torch.randn(B, S, generator=g)
    nz = torch.randn(B, 1, 2 * H, 2 * W, generator=g)
    with torch.no_grad():
        ref = R.styled_conv({k: v.double() for k, v in P.items()}, 'q', x.double(), wlat.double(), nz.double(), upsample=True)
        # the same style / demodulation factors the host mirror computes (EqualLinear + rsqrt of the squared-weight sums, model.py:236-241)
        s = R.equal_linear(wlat.double(), P['q.conv.modulation.weight'].double(), P['q.conv.modulation.bias'].double())
        w = P['q.conv.weight'][0].double()
        scale = 1.0 / math.sqrt(K * 9)
        d = torch.rsqrt(((scale * w[None] * s[:, None, :, None, None]) ** 2).sum([2, 3, 4]) + 1e-8)
    k1 = torch.tensor([1., 3., 3., 1.])
    k = (k1[:, None] * k1[None, :] / 64 * 4).contiguous().to(dev)
    wvb = ops.pack_upconv_vblur(P['q.conv.weight'][0].to(dev), scale, k)
    assert wvb is not None and ops.upconv_vblur_supported(B, K, M, H, W)
    xs = ops.to_sform(x.to(dev), s.float().to(dev))
    _lib.dispatch_reset()
    y = ops.upconv_vblur_fform(xs, wvb, out_scale=d.float().to(dev), bias=P['q.activate.bias'].to(dev), noise=nz.to(dev),
                               noise_weight=P['q.noise.weight'].to(dev), act=True)
    assert _lib.dispatch_count('upvb') == 1
    err = (y.to_nchw().double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-5, err
