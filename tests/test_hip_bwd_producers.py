"""Fused backward producers (csrc/bwd_producers.hip): the activation gradient written directly in the next matrix
kernel's input layout must be the SAME data as the two-pass path (act_bwd_fused -> to_sform / blurT_to_sform_phases)
when both use the same power-of-two range scale; the carried-scale W+ loop must follow the exact-scale loop."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oodgan import synth  # noqa: E402


def _same_sform(a, b, tol=5e-7):
    """S-form buffers as values hi+lo (records of 4 x 8 f16: hi, hi, lo, lo).  The fused kernel and the two-pass path
    may contract their fp32 expressions differently (1 ulp of fp32), so compare values, plus exact zeros in the same
    places (border / padding)."""
    va = a.data.view(-1, 4, 8).float()
    vb = b.data.view(-1, 4, 8).float()
    va, vb = va[:, :2] + va[:, 2:], vb[:, :2] + vb[:, 2:]
    assert torch.equal(va == 0, vb == 0)
    assert (va - vb).abs().max().item() <= tol * vb.abs().max().item()


def _inputs(B, C, H, W, seed, rgb, dev):
    t = lambda name, shape, *a: synth.normal(name, shape, seed, *a).to(dev)
    out = t('out', (B, C, H, W))
    g_feat = t('g', (B, C, H, W), 3e-4)
    noise = t('nz', (B, 1, H, W))
    nw = torch.tensor([0.1], device=dev)
    bias = t('bias', (C,), 0.1)
    d = t('d', (B, C), 0.2, 1.0).abs()
    kw = {}
    if rgb:
        kw = dict(g_rgb=t('grgb', (B, 3, H, W), 1e-3), w_rgb=t('wrgb', (3, C)), s_rgb=t('srgb', (B, C), 0.3, 1.0))
    return out, g_feat, noise, nw, bias, d, kw


@pytest.mark.parametrize('B,C,H,W,rgb', [(2, 32, 16, 16, True), (1, 48, 8, 20, False), (2, 16, 4, 4, True), (1, 24, 40, 36, True)])
def test_act_bwd_sform_matches_two_pass(B, C, H, W, rgb):
    from oodgan import ops
    dev = torch.device('cuda:0')
    out, g_feat, noise, nw, bias, d, kw = _inputs(B, C, H, W, 11 + C, rgb, dev)
    if rgb:
        g_pre, r0, t0, mul2 = ops.act_bwd_fused(out, g_feat, noise, nw, bias, kw['g_rgb'], kw['w_rgb'], kw['s_rgb'], want_scale=True, dscale=d)
    else:
        g_pre, r0, t0, mul2 = ops.act_bwd_fused(out, g_feat, noise, nw, bias, want_scale=True, dscale=d)
    ref = ops.to_sform(g_pre, d, mul2)
    dst = ops.SForm(B, C, H, W, dev)
    state = mul2.clone()
    r1, t1, part_m = ops.act_bwd_producer(out, g_feat, noise, nw, bias, d, state, dst, **kw)
    _same_sform(dst, ref)
    scale = max(1e-30, r0.abs().max().item())
    assert (r1 - r0).abs().max().item() <= 1e-5 * scale
    if rgb:
        assert (t1 - t0).abs().max().item() <= 1e-5 * max(1e-30, t0.abs().max().item())
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.absmax_scale_check(part_m, state, flag)
    assert flag.item() == 0 and torch.equal(state, mul2)        # same maximum -> same next scale, inside the window
    # a scale that is off by 2^20 must be reported
    bad = mul2 * torch.tensor([2.0 ** -20, 2.0 ** 20], device=dev)
    ops.absmax_scale_check(part_m, bad, flag)
    assert flag.item() == 1 and torch.equal(bad, mul2)


@pytest.mark.parametrize('B,C,H,W,rgb', [(2, 32, 8, 8, False), (1, 16, 4, 4, False), (1, 40, 18, 32, True)])
def test_act_bwd_blurT_phases_matches_two_pass(B, C, H, W, rgb):
    """H,W = input size of the up-conv; the activations are (2H,2W)."""
    from oodgan import ops
    from oracle import ref_cpu as R
    dev = torch.device('cuda:0')
    out, g_feat, noise, nw, bias, d, kw = _inputs(B, C, 2 * H, 2 * W, 5 + C, rgb, dev)
    k = torch.flip(R.make_kernel([1, 3, 3, 1]) * 4.0, [0, 1]).contiguous().to(dev)
    if rgb:
        g_pre, r0, t0, mul2 = ops.act_bwd_fused(out, g_feat, noise, nw, bias, kw['g_rgb'], kw['w_rgb'], kw['s_rgb'], want_scale=True, dscale=d)
    else:
        g_pre, r0, t0, mul2 = ops.act_bwd_fused(out, g_feat, noise, nw, bias, want_scale=True, dscale=d)
    ref = ops.blurT_to_sform_phases(g_pre, k, d, mul2)
    dst = ops.SFormPhases(B, C, H, W, dev)
    r1, t1, part_m = ops.act_bwd_producer(out, g_feat, noise, nw, bias, d, mul2.clone(), dst, blur_kernel=k, **kw)
    _same_sform(dst, ref)
    assert (r1 - r0).abs().max().item() <= 1e-5 * max(1e-30, r0.abs().max().item())
    if rgb:
        assert (t1 - t0).abs().max().item() <= 1e-5 * max(1e-30, t0.abs().max().item())
    state, flag = mul2.clone(), torch.zeros(1, dtype=torch.int32, device=dev)
    ops.absmax_scale_check(part_m, state, flag)
    assert flag.item() == 0 and torch.equal(state, mul2)


def test_wplus_loop_with_carried_scales_follows_exact_loop():
    """6 W+ steps with the fused producers (scale carried from step t-1) vs the per-step exact scales."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    dev = torch.device('cuda:0')
    size, B = 64, 2
    P = synth.generator_state(size, seed=5)
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
    target = synth.make_images(size, B, seed=9).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    w0 = synth.make_latents(size, B, seed=14).to(dev)
    assert eng.fused_bwd
    w1, l1 = WPlusInverter(eng).invert(target, w0, noises, steps=6)
    assert len(eng.bwd_state) == 2 * (6 - 2) + 1 and not eng.bwd_scale_violated()
    eng.fused_bwd = False
    w2, l2 = WPlusInverter(eng).invert(target, w0, noises, steps=6)
    eng.fused_bwd = True
    assert (l1 - l2).abs().max().item() <= 1e-5 * l2.abs().max().item()
    dw = (w1 - w2).abs()
    assert (dw < 1e-4).float().mean().item() > 0.999, dw.max().item()


def test_violated_scale_falls_back_to_exact_loop():
    from oodgan.engine import GeneratorEngine, WPlusInverter
    dev = torch.device('cuda:0')
    size, B = 32, 1
    P = synth.generator_state(size, seed=5)
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
    target = synth.make_images(size, B, seed=9).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    w0 = synth.make_latents(size, B, seed=14).to(dev)
    inv = WPlusInverter(eng, use_plan=False)      # the fault is injected through the Python-driven step (call counting)
    w_ref, l_ref = inv.invert(target, w0, noises, steps=4)
    # sabotage: after the first backward, blow one carried scale out of range
    orig, calls = eng.backward, {'n': 0}

    def sabotaged(gimg, gs=1.0, carry_scale=False):
        g = orig(gimg, gs, carry_scale=carry_scale)
        calls['n'] += 1
        if calls['n'] == 1 and eng.fused_bwd:
            next(iter(eng.bwd_state.values())).mul_(torch.tensor([2.0 ** 30, 2.0 ** -30], device=dev))
        return g
    eng.backward = sabotaged
    w, l = inv.invert(target, w0, noises, steps=4)
    del eng.backward
    assert calls['n'] == 8                                   # 4 flagged steps + the 4 steps of the exact re-run
    assert torch.isfinite(w).all() and (l - l_ref).abs().max().item() <= 1e-5 * l_ref.abs().max().item()


@pytest.mark.parametrize('B,Co,Ci,H,W,rgb', [(2, 32, 128, 16, 32, True), (1, 48, 64, 9, 40, True), (1, 64, 256, 8, 36, False),
                                             (2, 32, 64, 40, 72, True), (3, 20, 64, 19, 36, True), (1, 16, 64, 8, 32, False)])
def test_s2_conv_with_fused_activation_backward(B, Co, Ci, H, W, rgb, tunable):
    """The stride-2 input-gradient conv whose epilogue continues with the activation backward of the layer below
    (oodgan_actbwd_fuse) == the same conv followed by act_bwd_producer on its fp32 result: S-form gradient, the three
    per-channel sums, and the maxima that drive the carried range scale."""
    import math
    from oodgan import ops
    dev = torch.device('cuda:0')
    tunable('s2_big_min_items', 0)
    assert ops.s2_fuse_supported(B, Co, Ci, 2 * H + 1, 2 * W + 1)
    t = lambda n, shp, std=1.0, mean=0.0: synth.normal('fz.' + n, shp, 40 + Ci, std, mean).to(dev)
    out_below = t('out', (B, Ci, H, W))                      # saved activation of the conv layer below (= dotx)
    w = t('w', (Co, Ci, 3, 3), 1.0 / math.sqrt(Ci * 9))
    s_up = t('s', (B, Ci), 0.3, 1.0)
    d_up = t('dup', (B, Co), 0.3, 1.0)
    gz = t('gz', (B, Co, 2 * H + 1, 2 * W + 1), 3e-4)
    noise, nw, bias = t('nz', (B, 1, H, W)), torch.tensor([0.1], device=dev), t('bias', (Ci,), 0.1)
    d_below = t('d', (B, Ci), 0.2, 1.0).abs()
    kw = dict(g_rgb=t('grgb', (B, 3, H, W), 1e-3), w_rgb=t('wrgb', (3, Ci)), s_rgb=t('srgb', (B, Ci), 0.3, 1.0)) if rgb else {}
    wpk_t = ops.pack_conv3x3(w, 1.0, transpose=True, flip=False, precision='f16s')
    mul_up = torch.tensor([2.0 ** -10, 2.0 ** 10], device=dev)
    gp = ops.to_sform_phases(gz, H, W, d_up, mul_up)
    # two passes: conv -> fp32 g_feat -> producer
    g_feat, dot0 = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s_up, dotx=out_below, in_mul2=mul_up)
    _, _, _, state = ops.act_bwd_fused(out_below, g_feat, noise, nw, bias, kw.get('g_rgb'), kw.get('w_rgb'), kw.get('s_rgb'),
                                       want_scale=True, dscale=d_below)
    ref = ops.SForm(B, Ci, H, W, dev)
    r0, t0, pm0 = ops.act_bwd_producer(out_below, g_feat, noise, nw, bias, d_below, state, ref, **kw)
    # fused
    dst = ops.SForm(B, Ci, H, W, dev)
    fz = ops.ActBwdFusion(dst, noise, nw, bias, d_below, state, **kw)
    y, dot1 = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s_up, dotx=out_below, in_mul2=mul_up, fuse=fz, want_y=False)
    assert y is None
    a, b_ = dst.data.float().view(-1, 2, 16), ref.data.float().view(-1, 2, 16)
    va, vb = a[:, 0] + a[:, 1], b_[:, 0] + b_[:, 1]              # hi + lo
    assert (va - vb).abs().max().item() <= 2e-6 * vb.abs().max().item()
    assert (dot1 - dot0).abs().max().item() <= 1e-5 * dot0.abs().max().item()
    assert (fz.r - r0).abs().max().item() <= 1e-5 * max(1e-30, r0.abs().max().item())
    if rgb:
        assert (fz.t - t0).abs().max().item() <= 1e-5 * max(1e-30, t0.abs().max().item())
    assert abs(fz.part_m.max().item() - pm0.max().item()) <= 1e-5 * pm0.max().item()
    # the border of the S-form must stay zero (only the interior is written)
    full = dst.data.float().abs().sum().item()
    assert full > 0
    if Ci % 32 == 0:
        # round 4: the saved activation handed over ONLY as the S-form its producer wrote for the up-conv (x * scale as f16 pairs,
        # oodgan_conv_args.dotx_sform): the epilogue decodes (hi + lo) / scale — same results to the rounding of that round trip; and
        # the way back to NCHW for the two-pass fallback
        sc = t('sfs', (B, Ci), 0.4, 1.3) * 2.0 ** 7
        sc[:, ::5] *= -1.0                                       # styles may be negative
        saved = ops.SFormSaved(ops.to_sform(out_below, sc), sc)
        back = saved.to_nchw()
        assert (back - out_below).abs().max().item() <= 1e-6 * out_below.abs().max().item()
        dst2 = ops.SForm(B, Ci, H, W, dev)
        fz2 = ops.ActBwdFusion(dst2, noise, nw, bias, d_below, state, **kw)
        _, dot2 = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s_up, dotx=saved, in_mul2=mul_up, fuse=fz2, want_y=False)
        a2 = dst2.data.float().view(-1, 2, 16)
        va2 = a2[:, 0] + a2[:, 1]
        assert (va2 - vb).abs().max().item() <= 4e-6 * vb.abs().max().item()
        assert (dot2 - dot0).abs().max().item() <= 1e-5 * dot0.abs().max().item()
        assert (fz2.r - r0).abs().max().item() <= 1e-5 * max(1e-30, r0.abs().max().item())
        if rgb:
            assert (fz2.t - t0).abs().max().item() <= 1e-5 * max(1e-30, t0.abs().max().item())
        # without the fused epilogue the same object falls back to its NCHW copy
        g_feat2, dot3 = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s_up, dotx=saved, in_mul2=mul_up)
        assert (g_feat2 - g_feat).abs().max().item() <= 1e-6 * g_feat.abs().max().item() and (dot3 - dot0).abs().max().item() <= 1e-5 * dot0.abs().max().item()


@pytest.mark.parametrize('B,C,H,W,sep', [(1, 32, 32, 32, True), (2, 16, 40, 64, True), (1, 48, 37, 66, True), (1, 24, 64, 34, False), (1, 16, 33, 128, True)])
def test_act_bwd_blurT_strip_walk_matches_two_pass_and_tile_kernel(B, C, H, W, sep, tunable):
    """The strip-walking form of the blur^T producer (up-sampling layers with H, W >= 32 and no ToRGB branch) against the
    two-pass path (act_bwd_fused -> blurT_to_sform_phases) and against the tile kernel; ragged strips / segments, partial
    channel blocks, and a kernel that is not rank-1."""
    from oodgan import ops
    from oracle import ref_cpu as R
    dev = torch.device('cuda:0')
    out, g_feat, noise, nw, bias, d, kw = _inputs(B, C, 2 * H, 2 * W, 9 + C, False, dev)
    k = torch.flip(R.make_kernel([1, 3, 3, 1]) * 4.0, [0, 1]).contiguous()
    if not sep:
        k[1, 2] += 0.37
        k[3, 0] -= 0.11
    k = k.to(dev)
    g_pre, r0, t0, mul2 = ops.act_bwd_fused(out, g_feat, noise, nw, bias, want_scale=True, dscale=d)
    ref = ops.blurT_to_sform_phases(g_pre, k, d, mul2)
    res = {}
    for mode in ('1', '0'):
        tunable('blurt_strip', int(mode))
        dst = ops.SFormPhases(B, C, H, W, dev)
        dst.data.fill_(float('nan'))                 # every record of the (H+1) x (W+1) grid must be written
        r1, t1, part_m = ops.act_bwd_producer(out, g_feat, noise, nw, bias, d, mul2.clone(), dst, blur_kernel=k)
        res[mode] = (dst, r1, part_m)
    for mode, (dst, r1, part_m) in res.items():
        va = dst.data.view(-1, 4, 8).float()
        vb = ref.data.view(-1, 4, 8).float()
        va, vb = va[:, :2] + va[:, 2:], vb[:, :2] + vb[:, 2:]
        nz = vb != 0
        assert torch.isfinite(va[nz]).all(), mode
        assert (va[nz] - vb[nz]).abs().max().item() <= 1e-6 * vb.abs().max().item(), mode
        assert (r1 - r0).abs().max().item() <= 1e-5 * max(1e-30, r0.abs().max().item()), mode
        state, flag = mul2.clone(), torch.zeros(1, dtype=torch.int32, device=dev)
        ops.absmax_scale_check(part_m, state, flag)
        assert flag.item() == 0 and torch.equal(state, mul2), mode
    # both kernels write the same set of records (the NaN fill survives only in padding neither touches)
    a, b_ = res['1'][0].data, res['0'][0].data
    assert torch.equal(torch.isnan(a.float()), torch.isnan(b_.float()))


@pytest.mark.parametrize('B,Co,Ci,H,W', [(1, 64, 64, 64, 64), (2, 128, 64, 72, 96), (1, 32, 32, 64, 128), (2, 24, 20, 80, 64)])
@pytest.mark.parametrize('deferred', [False, True])
def test_s1_conv_with_act_gradient_of_dotx_then_blurT_producer(B, Co, Ci, H, W, deferred, tunable):
    """oodgan_conv_args.dot_actgrad: the stride-1 input-gradient conv above an up-sampling layer returns
    g_pre = dx * act'(out_below) (8-wave kernel for >= 64 channels, strip conv kernel for 17..32); the blur^T producer then
    runs on g_pre alone (out=None) and the layer's r sum is finished from its noise / bias term plus out_scale * dot.
    Against conv -> act_bwd_producer(out, dx): same g_pre bits, same phase-split S-form bits, same maxima, r to 1e-5."""
    import math
    from oodgan import ops
    from oracle import ref_cpu as R
    dev = torch.device('cuda:0')
    tunable('s1_big_min_items', 0)
    assert ops.s1_actgrad_supported(B, Co, Ci, H, W)
    t = lambda n, shp, std=1.0, mean=0.0: synth.normal('pre.' + n, shp, 60 + Ci + Co, std, mean).to(dev)
    out_below = t('out', (B, Ci, H, W))                      # output of the up-sampling layer below (= dotx)
    w = t('w', (Co, Ci, 3, 3), 1.0 / math.sqrt(Ci * 9))
    s2, d2 = t('s', (B, Ci), 0.3, 1.0), t('d2', (B, Co), 0.3, 1.0)
    g2 = t('g2', (B, Co, H, W), 3e-4)                        # g_pre of the conv layer above
    noise, nw, bias = t('nz', (B, 1, H, W)), torch.tensor([0.1], device=dev), t('bias', (Ci,), 0.1)
    d_below = t('d', (B, Ci), 0.2, 1.0).abs()
    k = torch.flip(R.make_kernel([1, 3, 3, 1]) * 4.0, [0, 1]).contiguous().to(dev)
    wpk_t = ops.pack_conv3x3(w, 1.0, transpose=True, flip=True, precision='f16s')
    mul_up = torch.tensor([2.0 ** -10, 2.0 ** 10], device=dev)
    gin = ops.to_sform(g2, d2, mul_up)
    # two passes
    dx, dot0 = ops.conv3x3(gin, wpk_t, Ci, ops.CONV_S1, out_scale=s2, dotx=out_below, in_mul2=mul_up)
    g_pre0, r0, _, state = ops.act_bwd_fused(out_below, dx, noise, nw, bias, want_scale=True, dscale=d_below)
    ref = ops.SFormPhases(B, Ci, H // 2, W // 2, dev)
    r0p, _, pm0 = ops.act_bwd_producer(out_below, dx, noise, nw, bias, d_below, state.clone(), ref, blur_kernel=k)
    # fused
    link = ops.DotActGrad()
    g_pre1, dot1 = ops.conv3x3(gin, wpk_t, Ci, ops.CONV_S1, out_scale=s2, dotx=out_below, in_mul2=mul_up, dot_actgrad=link)
    assert torch.equal(g_pre1, g_pre0)                       # the same fp32 product, only made in the conv's epilogue
    assert torch.equal(dot1, dot0)
    dst = ops.SFormPhases(B, Ci, H // 2, W // 2, dev)
    jobs = ops.BwdJobs() if deferred else None
    r1, tn, pm1 = ops.act_bwd_producer(None, g_pre1, noise, nw, bias, d_below, state.clone(), dst, blur_kernel=k, jobs=jobs, dot_of=link)
    if deferred:
        jobs.run(torch.zeros(1, dtype=torch.int32, device=dev))
    assert tn is None
    assert torch.equal(dst.data, ref.data)
    assert pm1.max().item() == pm0.max().item()
    tol = 1e-5 * max(1e-30, r0.abs().max().item())
    assert (r1 - r0).abs().max().item() <= tol and (r1 - r0p).abs().max().item() <= tol
    # the tile kernel has no such form: loud error, no silent fallback
    tunable('blurt_strip', int('0'))
    assert not ops.s1_actgrad_supported(B, Co, Ci, H, W)
    with pytest.raises(RuntimeError):
        ops.act_bwd_producer(None, g_pre1, noise, nw, bias, d_below, state.clone(), dst, blur_kernel=k, dot_of=link)


def test_fform_hand_off_of_the_last_styled_conv():
    """F-form (oodgan_conv_args.y_fform): the strip conv writes the same values as its NCHW form (bit for bit, through
    oodgan_from_fform), and oodgan_act_bwd_sform_f on it reproduces oodgan_act_bwd_sform on the NCHW tensor: S-form values and
    maxima to an fp32 ulp (same formula per element), the per-channel sums to rounding (different summation order)."""
    import math
    from oodgan import ops
    dev = torch.device('cuda:0')
    B, C, H, W = 2, 32, 48, 64
    x = synth.normal('ff.x', (B, C, H, W), 1)
    w = synth.normal('ff.w', (C, C, 3, 3), 2, 1.0 / math.sqrt(C * 9))
    s = synth.normal('ff.s', (B, C), 3, 0.3, 1.0).to(dev)
    d = synth.normal('ff.d', (B, C), 4, 0.3, 1.0).to(dev)
    nz = synth.normal('ff.nz', (B, 1, H, W), 5).to(dev)
    nw = torch.tensor([0.37], device=dev)
    bias = synth.normal('ff.b', (C,), 6).to(dev)
    wrgb = synth.normal('ff.wr', (3, C), 7).to(dev)
    srgb = synth.normal('ff.sr', (B, C), 8, 0.3, 1.0).to(dev)
    xs = ops.to_sform(x.to(dev), s)
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    kw = dict(out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU, rgb=(wrgb, srgb))
    y, rgb = ops.conv3x3(xs, wpk, C, ops.CONV_S1, **kw)
    yf, rgbf = ops.conv3x3(xs, wpk, C, ops.CONV_S1, y_fform=True, **kw)
    assert isinstance(yf, ops.FForm) and torch.equal(yf.to_nchw(), y) and torch.equal(rgbf, rgb)
    grgb = (1e-3 * synth.normal('ff.g', (B, 3, H, W), 9)).to(dev)
    mul2 = torch.tensor([2.0 ** -9, 2.0 ** 9], device=dev)
    dst_a, dst_b = ops.SForm(B, C, H, W, dev), ops.SForm(B, C, H, W, dev)
    ra, ta, ma = ops.act_bwd_producer(y, None, nz, nw, bias, d, mul2, dst_a, g_rgb=grgb, w_rgb=wrgb, s_rgb=srgb)
    rb, tb, mb = ops.act_bwd_producer(yf, None, nz, nw, bias, d, mul2, dst_b, g_rgb=grgb, w_rgb=wrgb, s_rgb=srgb)
    # same formula per element; the two kernels' FMA contraction differs: values (hi + lo) agree to an fp32 ulp
    rec = lambda t: (lambda r: torch.cat([r[:, 0] + r[:, 2], r[:, 1] + r[:, 3]], 1))(t.data.float().reshape(-1, 4, 8))
    va, vb = rec(dst_a), rec(dst_b)
    assert torch.equal(va != 0, vb != 0) and (va - vb).abs().max().item() <= 3e-7 * va.abs().max().item()
    assert abs(ma.max().item() - mb.max().item()) <= 3e-7 * ma.max().item()
    assert (ra - rb).abs().max().item() <= 1e-5 * ra.abs().max().item()
    assert (ta - tb).abs().max().item() <= 1e-5 * ta.abs().max().item()
