"""precision 'f16s-g2' (round 6; include/oodgan.h, oodgan_conv_args.x_hi_only): the input-gradient convs of the split-f16 path with the
BACK-PROPAGATED gradient rounded to f16 before each contraction — g_hi * (w_hi + w_lo), two matrix instructions per product instead of
three.  The reference has no counterpart (torch autograd runs conv2d's backward in fp32, model.py:233-274), so the checks are:

  * exactness of what the kernels claim to compute: on an operand that IS f16-representable the two-instruction instance must reproduce
    the three-instruction one (the dropped hi*lo term is zero), and on a general operand it must reproduce the three-instruction
    instance fed the f16-rounded operand — per kernel family (8-wave stride-1, 8-wave stride-2 plain and fused, F-form strip conv);
  * the engine's dL/dW+ against the oracle's float64 autograd at small sizes (the 1024² step against the reference's float64 generator
    is in test_hip_wplus_golden.py, the 100-step horizon in test_hip_wplus_long.py).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oodgan import synth  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _g2(wpk):
    wpk.x_hi_only = True
    return wpk


@pytest.mark.parametrize('pre', [False, True])
@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 128, 128, 32, 64), (1, 144, 192, 40, 33), (2, 256, 64, 17, 32)])
def test_s1_big_two_instruction_instance(dev, B, Ci, Co, H, W, pre, tunable):
    from oodgan import ops, _lib
    tunable('s1_big_min_items', 1)
    x = synth.normal('g2.x', (B, Ci, H, W), 1).to(dev)
    w = synth.normal('g2.w', (Co, Ci, 3, 3), 2, 1.0 / math.sqrt(Ci * 9)).to(dev)
    s = synth.normal('g2.s', (B, Ci), 3, 0.3, 1.0).to(dev)
    d = synth.normal('g2.d', (B, Co), 4, 0.3, 1.0).to(dev)
    dotx = synth.normal('g2.dx', (B, Co, H, W), 8).to(dev)
    mul2 = torch.tensor([2.0 ** -5, 2.0 ** 5], device=dev)
    w3, w2 = ops.pack_conv3x3(w, precision='f16s'), _g2(ops.pack_conv3x3(w, precision='f16s'))
    kw = lambda: dict(out_scale=d, dotx=dotx, in_mul2=mul2, dot_actgrad=ops.DotActGrad() if pre else None)
    xr = (x * s[:, :, None, None] * mul2[1]).half().float() / mul2[1]            # what the hi half of the S-form holds
    xs, xs_r = ops.to_sform(x, s, mul2), ops.to_sform(xr, None, mul2)
    _lib.dispatch_reset()
    y2, dot2 = ops.conv3x3(xs, w2, Co, ops.CONV_S1, **kw())
    assert _lib.dispatch_count('s1big_g2') == 1 and _lib.dispatch_count('s1big') == 1
    y3r, dot3r = ops.conv3x3(xs_r, w3, Co, ops.CONV_S1, **kw())                 # three instructions on the rounded operand (lo == 0)
    y2r, dot2r = ops.conv3x3(xs_r, w2, Co, ops.CONV_S1, **kw())                 # two instructions on it
    assert _lib.dispatch_count('s1big_g2') == 2
    assert _rel(y2, y3r) < 2e-6 and _rel(dot2, dot3r) < 2e-5
    assert _rel(y2r, y3r) < 2e-6 and _rel(dot2r, dot3r) < 2e-5
    # ... and the distance to the full-precision gradient is the f16 rounding of the operand: 2^-11 per element, averaged by the 9K-term sums
    y3, dot3 = ops.conv3x3(xs, w3, Co, ops.CONV_S1, **kw())
    e = _rel(y2, y3)
    print(f's1big g2 vs 3-instruction: max rel {e:.2e}, dot {_rel(dot2, dot3):.2e}')
    assert 1e-6 < e < 1e-3 and _rel(dot2, dot3) < 1e-3
    # a forward call (no dotx) ignores the flag
    _lib.dispatch_reset()
    yf2 = ops.conv3x3(xs, w2, Co, ops.CONV_S1, out_scale=d, in_mul2=mul2)
    yf3 = ops.conv3x3(xs, w3, Co, ops.CONV_S1, out_scale=d, in_mul2=mul2)
    assert _lib.dispatch_count('s1big_g2') == 0 and torch.equal(yf2, yf3)


@pytest.mark.parametrize('B,Co,Ci,H,W', [(2, 64, 128, 16, 32), (1, 48, 256, 9, 40), (2, 32, 64, 24, 33)])
def test_s2_big_two_instruction_instance(dev, B, Co, Ci, H, W, tunable):
    from oodgan import ops, _lib
    tunable('s2_big_min_items', 0)
    x = synth.normal('g2s.x', (B, Ci, H, W), 1).to(dev)
    w = synth.normal('g2s.w', (Co, Ci, 3, 3), 2, 1.0 / math.sqrt(Ci * 9)).to(dev)
    s = synth.normal('g2s.s', (B, Ci), 3, 0.3, 1.0).to(dev)
    d = synth.normal('g2s.d', (B, Co), 4, 0.3, 1.0).to(dev)
    gz = synth.normal('g2s.gz', (B, Co, 2 * H + 1, 2 * W + 1), 5).to(dev)
    mul2 = torch.tensor([2.0 ** -4, 2.0 ** 4], device=dev)
    w3 = ops.pack_conv3x3(w, 1.0, transpose=True, flip=False, precision='f16s')
    w2 = _g2(ops.pack_conv3x3(w, 1.0, transpose=True, flip=False, precision='f16s'))
    gr = (gz * d[:, :, None, None] * mul2[1]).half().float() / mul2[1]
    gp, gp_r = ops.to_sform_phases(gz, H, W, d, mul2), ops.to_sform_phases(gr, H, W, None, mul2)
    _lib.dispatch_reset()
    dx2, dot2 = ops.conv3x3(gp, w2, Ci, ops.CONV_S2, out_scale=s, dotx=x, in_mul2=mul2)
    assert _lib.dispatch_count('s2big_g2') == 1
    dx3r, dot3r = ops.conv3x3(gp_r, w3, Ci, ops.CONV_S2, out_scale=s, dotx=x, in_mul2=mul2)
    assert _rel(dx2, dx3r) < 2e-6 and _rel(dot2, dot3r) < 2e-5
    dx3, dot3 = ops.conv3x3(gp, w3, Ci, ops.CONV_S2, out_scale=s, dotx=x, in_mul2=mul2)
    e = _rel(dx2, dx3)
    print(f's2big g2 vs 3-instruction: max rel {e:.2e}, dot {_rel(dot2, dot3):.2e}')
    assert 1e-6 < e < 1e-3 and _rel(dot2, dot3) < 1e-3


@pytest.mark.parametrize('B,Co,Ci,H,W,rgb', [(2, 32, 128, 16, 32, True), (1, 64, 256, 8, 36, False), (2, 32, 64, 40, 72, True)])
def test_s2_big_fused_epilogue_two_instruction_instance(dev, B, Co, Ci, H, W, rgb, tunable):
    """The fused activation backward behind the two-instruction K loop: same S-form gradient / sums as behind the three-instruction
    loop on the f16-rounded operand."""
    from oodgan import ops, _lib
    tunable('s2_big_min_items', 0)
    t = lambda n, shp, std=1.0, mean=0.0: synth.normal('g2f.' + n, shp, 40 + Ci, std, mean).to(dev)
    out_below = t('out', (B, Ci, H, W))
    w = t('w', (Co, Ci, 3, 3), 1.0 / math.sqrt(Ci * 9))
    s_up, d_up = t('s', (B, Ci), 0.3, 1.0), t('dup', (B, Co), 0.3, 1.0)
    gz = t('gz', (B, Co, 2 * H + 1, 2 * W + 1), 3e-4)
    noise, nw, bias = t('nz', (B, 1, H, W)), torch.tensor([0.1], device=dev), t('bias', (Ci,), 0.1)
    d_below = t('d', (B, Ci), 0.2, 1.0).abs()
    kw = dict(g_rgb=t('grgb', (B, 3, H, W), 1e-3), w_rgb=t('wrgb', (3, Ci)), s_rgb=t('srgb', (B, Ci), 0.3, 1.0)) if rgb else {}
    w3 = ops.pack_conv3x3(w, 1.0, transpose=True, flip=False, precision='f16s')
    w2 = _g2(ops.pack_conv3x3(w, 1.0, transpose=True, flip=False, precision='f16s'))
    mul_up = torch.tensor([2.0 ** -10, 2.0 ** 10], device=dev)
    gr = (gz * d_up[:, :, None, None] * mul_up[1]).half().float() / mul_up[1]
    gp, gp_r = ops.to_sform_phases(gz, H, W, d_up, mul_up), ops.to_sform_phases(gr, H, W, None, mul_up)
    g_feat, _ = ops.conv3x3(gp_r, w3, Ci, ops.CONV_S2, out_scale=s_up, dotx=out_below, in_mul2=mul_up)
    _, _, _, state = ops.act_bwd_fused(out_below, g_feat, noise, nw, bias, kw.get('g_rgb'), kw.get('w_rgb'), kw.get('s_rgb'), want_scale=True, dscale=d_below)
    res = []
    for gin, wpk in ((gp_r, w3), (gp, w2)):
        dst = ops.SForm(B, Ci, H, W, dev)
        fz = ops.ActBwdFusion(dst, noise, nw, bias, d_below, state, **kw)
        _lib.dispatch_reset()
        _, dot = ops.conv3x3(gin, wpk, Ci, ops.CONV_S2, out_scale=s_up, dotx=out_below, in_mul2=mul_up, fuse=fz, want_y=False)
        assert _lib.dispatch_count('s2big_fuse') == 1 and _lib.dispatch_count('s2big_g2') == (1 if wpk is w2 else 0)
        a = dst.data.float().view(-1, 2, 16)
        res.append((a[:, 0] + a[:, 1], dot, fz.r, fz.t if rgb else None, fz.part_m.max()))
    (v0, dot0, r0, t0, m0), (v1, dot1, r1, t1, m1) = res
    assert _rel(v1, v0) < 4e-6 and _rel(dot1, dot0) < 2e-5 and _rel(r1, r0) < 2e-5
    if rgb:
        assert _rel(t1, t0) < 2e-5
    assert abs(float(m1) - float(m0)) <= 1e-5 * float(m0)


def _to_fform(x):
    B, C, H, W = x.shape
    return x.view(B, C // 16, 16, H, W).permute(0, 1, 3, 4, 2).contiguous().view(B, C, H, W)


@pytest.mark.parametrize('B,H,W', [(2, 64, 64), (1, 40, 96), (1, 512, 512)])
@pytest.mark.parametrize('pre', [True, False])
def test_fform_strip_conv_two_instruction_instance(dev, B, H, W, pre):
    """conv_f16s_stripx (input gradient with the activation backward inside): the operand is produced in the kernel, so the check is
    against the three-instruction instance with the rounding of the operand as the tolerance — the sums the conversion forms on the way
    (r, t, the recorded maximum) do not depend on the operand's format and must agree exactly."""
    from oodgan import ops, _lib
    g = torch.Generator().manual_seed(H * 5 + W + 3)
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
    w3 = ops.pack_conv3x3(w, transpose=True, flip=True, precision='f16s')
    w2 = _g2(ops.pack_conv3x3(w, transpose=True, flip=True, precision='f16s'))
    mul2 = torch.tensor([2.0 ** -14, 2.0 ** 14], device=dev)
    o2f, o1f = ops.FForm(_to_fform(out2)), ops.FForm(_to_fform(out1))
    xa3 = ops.ActBwdX(nz, nw, bias, d, mul2, g_rgb, w_rgb, s_rgb)
    dx3, dot3 = ops.conv3x3(o2f, w3, C, ops.CONV_S1, out_scale=s1, dotx=o1f, in_mul2=mul2, dot_actgrad=ops.DotActGrad() if pre else None, xf_act=xa3)
    _lib.dispatch_reset()
    xa2 = ops.ActBwdX(nz, nw, bias, d, mul2, g_rgb, w_rgb, s_rgb)
    dx2, dot2 = ops.conv3x3(o2f, w2, C, ops.CONV_S1, out_scale=s1, dotx=o1f, in_mul2=mul2, dot_actgrad=ops.DotActGrad() if pre else None, xf_act=xa2)
    assert _lib.dispatch_count('stripx_g2') == 1 and _lib.dispatch_count('stripx') == 1
    e = _rel(dx2, dx3)
    rms = float((dx2 - dx3).pow(2).mean().sqrt() / dx3.pow(2).mean().sqrt())
    print(f'stripx g2 vs 3-instruction: max rel {e:.2e}, rms rel {rms:.2e}, dot {_rel(dot2, dot3):.2e}')
    assert 1e-6 < e < 1e-3 and rms < 3e-4 and _rel(dot2, dot3) < 1e-3
    assert torch.equal(xa2.r, xa3.r) and torch.equal(xa2.t, xa3.t) and torch.equal(xa2.part_m, xa3.part_m)
    # the same operand through the two-pass path: S-form producer, its records rounded to their hi half, then the S-form strip kernel
    gin = ops.SForm(B, C, H, W, dev)
    ops.act_bwd_producer(o2f, None, nz, nw, bias, d, mul2, gin, g_rgb=g_rgb, w_rgb=w_rgb, s_rgb=s_rgb)
    rec = gin.data.view(-1, 2, 16)
    rec[:, 1].zero_()                                   # lo halves := 0 — the producer's hi half is RNE(f16) of the same fp32 value
    dxr, dotr = ops.conv3x3(gin, w3, C, ops.CONV_S1, out_scale=s1, dotx=out1, in_mul2=mul2, dot_actgrad=ops.DotActGrad() if pre else None)
    # (one fp32 rounding apart from the producer — s_rgb folded into the slope — so a handful of operands round to the neighbouring f16;
    # ONE such operand moves the outputs it feeds by 2^-11 |g w|, i.e. ~1e-4 of max|dx|: the comparison is on the rms and on the share of
    # outputs touched)
    err = (dx2 - dxr).abs()
    rms_r = float(err.pow(2).mean().sqrt() / dxr.pow(2).mean().sqrt())
    touched = float((err > 1e-5 * dxr.abs().max()).float().mean())
    print(f'stripx g2 vs S-form strip kernel on the hi halves: rms rel {rms_r:.2e}, outputs off by > 1e-5 max: {touched:.2e}, max rel {_rel(dx2, dxr):.2e}')
    assert rms_r < 2e-5 and touched < 5e-3 and _rel(dx2, dxr) < 1e-3 and _rel(dot2, dotr) < 5e-5


@pytest.mark.parametrize('size,B', [(16, 2), (64, 1), (128, 2)])
def test_generator_backward_g2_vs_oracle_autograd(dev, size, B, tunable):
    """dL/dW+ of the engine with precision='f16s-g2' vs the oracle's float64 autograd (every 8-wave kernel reachable at these sizes is
    forced by the item thresholds), and beside it the three-instruction engine on the same inputs."""
    from oodgan.engine import GeneratorEngine
    from oodgan import ops, _lib
    for name in ('s1_big_min_items', 's2_big_min_items', 't2_big_min_items'):
        tunable(name, 1)
    P = synth.generator_state(size, seed=5)
    lat = synth.make_latents(size, B, seed=14)
    noises = synth.make_noises(size, B, seed=7)
    target = synth.make_images(size, B, seed=9)
    w = lat.double().requires_grad_(True)
    img_ref = R.generator_forward({k: v.double() for k, v in P.items()}, w, [n.double() for n in noises], size)
    R.wplus_loss(img_ref, target.double()).backward()
    gref = w.grad
    gmul = ops.loss_scale_for(3 * size * size)
    rels = {}
    for prec in ('f16s', 'f16s-g2'):
        eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size, precision=prec)
        eng.reset_bwd_state()
        eng.reset_fwd_state()
        for rep in range(2):                # exact scales, then carried scales + fused producers
            _lib.dispatch_reset()
            img = eng.forward(lat.to(dev), [n.to(dev) for n in noises], save=True, range_mode='carry')
            loss, gimg = ops.mse_loss_grad(img, target.to(dev), gmul)
            glat = eng.backward(gimg, gmul, carry_scale=True)
            n_g2 = sum(_lib.dispatch_count(k) for k in ('s1big_g2', 's2big_g2', 'stripx_g2'))
            assert (n_g2 > 0) if (prec == 'f16s-g2' and size >= 64) else (prec == 'f16s-g2' or n_g2 == 0), (prec, size, n_g2)
            rels[prec, rep] = float((glat.double().cpu() - gref).abs().max() / gref.abs().max())
        assert not eng.bwd_scale_violated() and not eng.fwd_range_violated()
    print(f'size {size}: dL/dw rel err vs f64 autograd: {rels}')
    assert max(rels.values()) < 3e-4


@pytest.mark.parametrize('pre', [False, True])
@pytest.mark.parametrize('B,Co,Ci,H,W', [(2, 64, 128, 32, 32), (1, 128, 64, 40, 72), (2, 32, 64, 64, 64)])
def test_hi_only_gradient_records_between_blurT_producer_and_s2_conv(dev, B, Co, Ci, H, W, pre, tunable):
    """oodgan_act_bwd_blurT_sform_phases_hi writes 32-byte hi-only records, the two-instruction 8-wave stride-2 conv reads them with
    oodgan_conv_args.x_hi_only = 2: the SAME hi halves as in the 64-byte records, so the conv's results are bit-identical to the full-record
    path (which ignores its lo halves) — with and without the fused activation backward; a conv that needs the lo halves refuses the buffer."""
    from oodgan import ops, _lib
    tunable('s2_big_min_items', 0)
    assert ops.blurT_hi_supported(H, W) and ops.s2_fuse_supported(B, Co, Ci, 2 * H + 1, 2 * W + 1)
    t = lambda n, shp, std=1.0, mean=0.0: synth.normal('hr.' + n, shp, 50 + Ci, std, mean).to(dev)
    k4 = torch.flip(synth.make_kernel() * 4.0, [0, 1]).contiguous().to(dev)
    out_up = t('outup', (B, Co, 2 * H, 2 * W))                   # saved output of the up-conv layer (its activation backward is the producer's job)
    g_feat = t('g', (B, Co, 2 * H, 2 * W), 1e-3)
    nz, nw, bias = t('nz', (B, 1, 2 * H, 2 * W)), torch.tensor([0.1], device=dev), t('bias', (Co,), 0.1)
    d_up = t('d', (B, Co), 0.2, 1.0).abs()
    mul2 = torch.tensor([2.0 ** -9, 2.0 ** 9], device=dev)
    x_in = t('xin', (B, Ci, H, W))                               # input of the up-conv (dotx of its input-gradient conv)
    s_in = t('s', (B, Ci), 0.3, 1.0)
    w = t('w', (Co, Ci, 3, 3), 1.0 / math.sqrt(Ci * 9))
    w2 = _g2(ops.pack_conv3x3(w, 1.0, transpose=True, flip=False, precision='f16s'))
    w3 = ops.pack_conv3x3(w, 1.0, transpose=True, flip=False, precision='f16s')
    res = []
    for hi in (False, True):
        gin = ops.SFormPhases(B, Co, H, W, dev)
        if pre:
            dag = ops.DotActGrad()
            dag.dot_part, dag.scale = torch.zeros(B, Co, 4, device=dev), d_up
            g_pre = g_feat * torch.where(out_up > 0, 2 ** 0.5, 0.2 * 2 ** 0.5)
            r, _, pm = ops.act_bwd_producer(None, g_pre, nz, nw, bias, d_up, mul2, gin, blur_kernel=k4, dot_of=dag, hi_only=hi)
        else:
            r, _, pm = ops.act_bwd_producer(out_up, g_feat, nz, nw, bias, d_up, mul2, gin, blur_kernel=k4, hi_only=hi)
        assert gin.hi_only == hi
        _lib.dispatch_reset()
        dx, dot = ops.conv3x3(gin, w2, Ci, ops.CONV_S2, out_scale=s_in, dotx=x_in, in_mul2=mul2)
        assert _lib.dispatch_count('s2big_g2') == 1 and _lib.dispatch_count('s2big_xh') == (1 if hi else 0)
        res.append((dx, dot, r, pm, gin))
    (dx0, dot0, r0, pm0, _), (dx1, dot1, r1, pm1, gin_hi) = res
    assert torch.equal(dx1, dx0) and torch.equal(dot1, dot0) and torch.equal(r1, r0) and torch.equal(pm1, pm0)
    if Ci % 32 == 0:
        # ... and behind the fused activation backward of the conv layer below
        out_below = t('ob', (B, Ci, H, W))
        d_below = t('db', (B, Ci), 0.2, 1.0).abs()
        _, _, _, state = ops.act_bwd_fused(out_below, dx0, nz[:, :, :H, :W].contiguous(), nw, t('b2', (Ci,), 0.1), want_scale=True, dscale=d_below)
        # the fused epilogue's own S-form output as hi-only records too (oodgan_actbwd_fuse.ys_hi_only), read by the two-instruction 8-wave
        # stride-1 conv (x_hi_only = 2): bit-identical to the full-record chain
        tunable('s1_big_min_items', 1)
        w1 = t('w1', (Ci, Ci, 3, 3), 1.0 / math.sqrt(Ci * 9))
        w1b = _g2(ops.pack_conv3x3(w1, transpose=True, flip=True, precision='f16s'))
        dot1x = t('d1x', (B, Ci, H, W))
        s1v = t('s1v', (B, Ci), 0.3, 1.0)
        outs = []
        for gin, hi in ((res[0][4], False), (gin_hi, True)):
            dst = ops.SForm(B, Ci, H, W, dev)
            dst.hi_only = hi
            fz = ops.ActBwdFusion(dst, nz[:, :, :H, :W].contiguous(), nw, t('b2', (Ci,), 0.1), d_below, state, hi_only=hi)
            _, dotf = ops.conv3x3(gin, w2, Ci, ops.CONV_S2, out_scale=s_in, dotx=out_below, in_mul2=mul2, fuse=fz, want_y=False)
            xh_ok = ops.s1_xh_supported(B, Ci, Ci, H, W)
            if hi and not xh_ok:
                continue
            _lib.dispatch_reset()
            dx1, dot1 = ops.conv3x3(dst, w1b, Ci, ops.CONV_S1, out_scale=s1v, dotx=dot1x, in_mul2=state)
            assert _lib.dispatch_count('s1big_xh') == (1 if hi else 0)
            outs.append((dx1, dot1, dotf, fz.r, fz.part_m.max()))
        if len(outs) == 2:
            for u, v in zip(outs[0], outs[1]):
                assert torch.equal(u, v)
    # a conv that reads the lo halves must refuse hi-only records
    with pytest.raises((RuntimeError, AssertionError)):
        ops.conv3x3(gin_hi, w3, Ci, ops.CONV_S2, out_scale=s_in, dotx=x_in, in_mul2=mul2)
