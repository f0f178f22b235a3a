"""GPU parity of the SAMM/SAIM ops and the full OOD forward (post-encoder) at 1024²."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oodgan import synth  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def close(a, b, tol=1e-4):
    a = a.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert err <= tol * ref, f'max abs err {err:.3e} (ref max {ref:.3e})'


def test_instance_norm_and_small_convs(dev):
    from oodgan import samm
    x = synth.normal('in.x', (2, 5, 13, 17), 1, 2.0, 0.7)
    g, b = synth.normal('in.g', (5,), 2, 0.1, 1.0), synth.normal('in.b', (5,), 3, 0.1)
    close(samm.instance_norm(x.to(dev), g.to(dev), b.to(dev)), R.instance_norm(x, g, b))
    close(samm.instance_norm(x.to(dev)), R.instance_norm(x))
    w = synth.normal('c1.w', (7, 5, 1, 1), 4)
    bb = synth.normal('c1.b', (7,), 5)
    close(samm.conv1x1(x.to(dev), w.to(dev), bb.to(dev)), F.conv2d(x, w, bb))
    x2 = synth.normal('c1.x2', (1, 130, 9, 9), 6)
    w2 = synth.normal('c1.w2', (35, 130, 1, 1), 7, 0.1)
    close(samm.conv1x1(x2.to(dev), w2.to(dev)), F.conv2d(x2, w2))
    # the wide form (64 channels per block; grids of >= 1024 blocks): ragged pixel count, partial channel block, K chunk remainder
    x3 = synth.normal('c1.x3', (2, 70, 130, 131), 10)
    w3w = synth.normal('c1.w3', (72, 70, 1, 1), 11, 0.1)
    b3 = synth.normal('c1.b3', (72,), 12)
    close(samm.conv1x1(x3.to(dev), w3w.to(dev), b3.to(dev)), F.conv2d(x3, w3w, b3), 2e-6)
    w3 = synth.normal('c3.w', (3, 5, 3, 3), 8, 0.3)
    sl = synth.normal('c3.s', (3,), 9, 0.05, 0.25)
    close(samm.conv3x3_small(x.to(dev), w3.to(dev), slope=sl.to(dev)), F.prelu(F.conv2d(x, w3, padding=1), sl))


def test_se_gate_vs_aten(dev):
    """SEModule (e4e/encoders/helpers.py:60-76): sigmoid(fc2(relu(fc1(avgpool(x)))))."""
    from oodgan import samm
    for (B, C, r, H) in ((3, 64, 16, 12), (1, 512, 16, 5), (2, 256, 16, 9)):
        x = synth.normal('se.x', (B, C, H, H + 1), 1, 1.0, 0.3)
        w1 = synth.normal('se.w1', (C // r, C, 1, 1), 2, 0.2)
        w2 = synth.normal('se.w2', (C, C // r, 1, 1), 3, 0.3)
        ref = torch.sigmoid(F.conv2d(torch.relu(F.conv2d(x.mean(dim=(2, 3), keepdim=True), w1)), w2)).reshape(B, C)
        g = samm.se_gate(samm.instnorm_stats(x.to(dev)), w1.to(dev), w2.to(dev))
        close(g, ref, 2e-6)


def test_conv3x3_fewout_vs_aten(dev):
    """The many-to-few 3x3 conv (AlignNet head, 2C -> 3) incl. in-bounds-only shift, PReLU, ragged K / sizes and the K split."""
    from oodgan import samm
    for (B, K, H, W, M) in ((2, 200, 20, 45, 3), (1, 1024, 32, 32, 3), (3, 16, 9, 7, 4), (1, 256, 64, 64, 1)):
        x = synth.normal('fo.x', (B, K, H, W), 1, 1.3, 0.1)
        w = synth.normal('fo.w', (M, K, 3, 3), 2, 0.05)
        sc = synth.normal('fo.sc', (B, K), 3, 0.2, 1.0)
        sh = synth.normal('fo.sh', (B, K), 4, 0.3)
        sl = synth.normal('fo.sl', (M,), 5, 0.05, 0.25)
        ref = F.prelu(torch.cat([F.conv2d(x[b:b + 1] * sc[b].view(1, K, 1, 1) + sh[b].view(1, K, 1, 1), w, padding=1) for b in range(B)]), sl)
        y = samm.conv3x3_fewout(x.to(dev), w.to(dev), sc.to(dev), sh.to(dev), slope=sl.to(dev))
        close(y, ref, 2e-5)
        close(samm.conv3x3_fewout(x.to(dev), w.to(dev)), F.conv2d(x, w, padding=1), 2e-5)
        assert torch.equal(y, samm.conv3x3_fewout(x.to(dev), w.to(dev), sc.to(dev), sh.to(dev), slope=sl.to(dev)))    # deterministic


def test_conv3x3_fewout2_with_shortcut_conv_vs_aten(dev):
    """Round 4: the AlignNet head conv from the transposed weight copy, together with the 1x1 shortcut conv of the same bottleneck
    (bottleneck_IR(2C, 3): res_layer[1] on the normalised input, shortcut_layer[0] on the raw one, e4e helpers.py:426-448) in one pass."""
    from oodgan import samm
    for (B, K, H, W, M, M2) in ((2, 200, 20, 45, 3, 3), (1, 1024, 32, 32, 3, 3), (3, 16, 9, 7, 4, 2), (1, 256, 64, 64, 1, 0), (8, 256, 40, 33, 3, 3), (2, 64, 20, 44, 3, 3), (2, 72, 13, 36, 3, 0), (1, 512, 128, 128, 3, 3)):
        x = synth.normal('f2.x', (B, K, H, W), 1, 1.3, 0.1)
        w = synth.normal('f2.w', (M, K, 3, 3), 2, 0.05)
        w11 = synth.normal('f2.w11', (M2, K, 1, 1), 6, 0.05) if M2 else None
        sc = synth.normal('f2.sc', (B, K), 3, 0.2, 1.0)
        sh = synth.normal('f2.sh', (B, K), 4, 0.3)
        sl = synth.normal('f2.sl', (M,), 5, 0.05, 0.25)
        ref = F.prelu(torch.cat([F.conv2d(x[b:b + 1] * sc[b].view(1, K, 1, 1) + sh[b].view(1, K, 1, 1), w, padding=1) for b in range(B)]), sl)
        wt, w11t = samm.fewout_weights(w.to(dev), None if w11 is None else w11.to(dev))
        y, y2 = samm.conv3x3_fewout2(x.to(dev), wt, M, sc.to(dev), sh.to(dev), slope=sl.to(dev), w11t=w11t, M2=M2)
        close(y, ref, 2e-5)
        if M2:
            close(y2, F.conv2d(x, w11), 2e-5)
        else:
            assert y2 is None
        ya, yb = samm.conv3x3_fewout2(x.to(dev), wt, M, sc.to(dev), sh.to(dev), slope=sl.to(dev), w11t=w11t, M2=M2)
        assert torch.equal(y, ya) and (y2 is None or torch.equal(y2, yb))          # deterministic
        close(samm.conv3x3_fewout2(x.to(dev), wt, M)[0], F.conv2d(x, w, padding=1), 2e-5)


def test_affine_apply_stats_equals_two_passes(dev):
    """y = x*sc + sh + res and the InstanceNorm statistics of y from one kernel: bit-identical to affine_apply + instnorm_stats."""
    from oodgan import samm
    for (B, C, H, W, res) in ((2, 24, 16, 16, True), (1, 7, 9, 13, True), (3, 16, 64, 64, False), (1, 4, 256, 256, True)):
        x = synth.normal('as.x', (B, C, H, W), 1, 1.7, 0.4).to(dev)
        r = synth.normal('as.r', (B, C, H, W), 2).to(dev) if res else None
        sc = synth.normal('as.sc', (B, C), 3, 0.2, 1.0).to(dev)
        sh = synth.normal('as.sh', (B, C), 4, 0.3).to(dev)
        y0 = samm.affine_apply(x, sc, sh, r)
        st0 = samm.instnorm_stats(y0)
        y1, st1 = samm.affine_apply_stats(x, sc, sh, r)
        assert torch.equal(y0, y1) and torch.equal(st0, st1)


def test_align_input_stats_equals_two_passes(dev):
    """cat[IN(gen) - IN(enc), IN(enc)] (SAMM/helpers.py:96-104) and the statistics of the result from one kernel: bit-identical."""
    from oodgan import samm
    for (B, C, H, W) in ((2, 24, 16, 16), (1, 7, 9, 13), (3, 16, 64, 64), (1, 4, 256, 256)):
        g = synth.normal('ai.g', (B, C, H, W), 1, 1.7, 0.4).to(dev)
        e = synth.normal('ai.e', (B, C, H, W), 2, 0.6, -0.2).to(dev)
        sg, se = samm.instnorm_stats(g), samm.instnorm_stats(e)
        a0 = samm.align_input(g, e, sg, se)
        st0 = samm.instnorm_stats(a0)
        a1, st1 = samm.align_input_stats(g, e, sg, se)
        assert torch.equal(a0, a1) and torch.equal(st0, st1)


def test_warp_blend_channel_chunks(dev):
    """grid_sample + lerp with channel counts that are not a multiple of the kernel's channel chunk, B > 1."""
    from oodgan import samm
    for (B, C, H, W) in ((2, 13, 24, 40), (3, 512, 32, 32), (1, 5, 7, 9)):
        t = synth.normal('wc.t', (B, C, H, W), 1)
        f = torch.cat([synth.normal('wc.f', (B, 2, H, W), 2, 0.4), synth.uniform('wc.a', (B, 1, H, W), 3)], 1)
        close(samm.warp_blend(t.to(dev), f.to(dev)), R.warp_blend(t, f), 1e-5)


def test_resize_index_math_exact(dev):
    """nearest / bilinear source-index math must match ATen bit for bit (mask indexing)."""
    from oodgan import samm
    for s in (32, 64, 128, 256, 37):
        x = synth.uniform('rs.x', (2, 1, s, s), s)
        yn = samm.resize_nearest(x.to(dev), 1024).cpu()
        assert torch.equal(yn, F.interpolate(x, size=1024)), f'nearest {s}->1024 not exact'
        yb = samm.resize_bilinear(x.to(dev), 1024).cpu()
        assert (yb - F.interpolate(x, size=(1024, 1024), mode='bilinear')).abs().max() < 2e-6
    x = synth.uniform('rs.y', (1, 3, 1024, 1024), 5)
    close(samm.resize_bilinear(x.to(dev), 256), F.interpolate(x, (256, 256), mode='bilinear'), 1e-5)


def test_field_ops_and_warp(dev, golden):
    from oodgan import samm
    g = golden('samm.npz')
    close(samm.field_add(g['warp_field'].to(dev), g['warp_field_prev'].to(dev), 0.08),
          R.spm_add(g['warp_field'], g['warp_field_prev'], 0.08), 1e-5)
    close(samm.field_upsample_add(g['prev'].to(dev), g['warp_field'].to(dev)),
          R.spm_upsample_add(g['prev'], g['warp_field']), 1e-5)
    close(samm.warp_blend(g['tgt'].to(dev), g['warp_field'].to(dev)), R.warp_blend(g['tgt'], g['warp_field']), 1e-5)
    # larger, non-square, displacement beyond the border (zeros padding)
    t = synth.normal('wb.t', (2, 6, 40, 56), 1)
    f = torch.cat([synth.normal('wb.f', (2, 2, 40, 56), 2, 0.3), synth.uniform('wb.a', (2, 1, 40, 56), 3)], 1)
    close(samm.warp_blend(t.to(dev), f.to(dev)), R.warp_blend(t, f), 1e-5)
    # bicubic(align_corners=True) up-sampling inside upsample_add at the real level ratios
    for hp, h in [(32, 64), (64, 128), (8, 16)]:
        prev = torch.cat([synth.normal('ua.d', (1, 2, hp, hp), 4, 0.05), synth.uniform('ua.a', (1, 1, hp, hp), 5)], 1)
        cur = torch.cat([synth.normal('ua.d2', (1, 2, h, h), 6, 0.05), synth.uniform('ua.a2', (1, 1, h, h), 7)], 1)
        close(samm.field_upsample_add(prev.to(dev), cur.to(dev)), R.spm_upsample_add(prev, cur), 1e-5)


def test_alignnet_and_spm_warp_vs_golden(dev, golden):
    from oodgan import samm
    g = golden('samm.npz')
    sd = synth.samm_state(8, 'm', seed=21)
    blk = samm.StyledscaleNshfitBlock(8, 8, 512, scale=0.08, cycle_align=2, diff_fAndg=True)
    res = blk.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    blk = blk.to(dev)
    src, tgt, prev = g['src'].to(dev), g['tgt'].to(dev), g['prev'].to(dev)
    close(blk.alignment.body(tgt, src), g['alignnet'], 2e-4)
    y, f = blk(src, None, image=tgt, aligned=None)
    close(y, g['warp_out'], 3e-4)
    close(f, g['warp_field'], 3e-4)
    y, f = blk(src, None, image=tgt, aligned=prev)
    close(y, g['warp_out_prev'], 3e-4)
    close(f, g['warp_field_prev'], 3e-4)


def test_alignnet_without_the_difference_input_vs_golden(dev, golden):
    """diff_fAndg=False (reference SAMM/helpers.py:98-101; round 5): AlignNet sees cat([IN(source), IN(target)]).  Vectors of the real
    SPM_Warp(diff_fAndg=False) on gold_samm's weights and inputs; fused-statistics and plain paths."""
    from oodgan import samm
    g, g0 = golden('samm_nodiff.npz'), golden('samm.npz')
    sd = synth.samm_state(8, 'm', seed=21)
    blk = samm.StyledscaleNshfitBlock(8, 8, 512, scale=0.08, cycle_align=2, diff_fAndg=False)
    blk.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    blk = blk.to(dev)
    src, tgt, prev = g0['src'].to(dev), g0['tgt'].to(dev), g0['prev'].to(dev)
    for fuse in (True, False):
        old = samm.FUSE_STATS
        try:
            samm.FUSE_STATS = fuse
            close(blk.alignment.body(tgt, src), g['alignnet'], 2e-4)
            y, f = blk(src, None, image=tgt, aligned=prev)
        finally:
            samm.FUSE_STATS = old
        close(y, g['warp_out_prev'], 3e-4)
        close(f, g['warp_field_prev'], 3e-4)


def test_mod_btn_feature_extractors_vs_reference(dev, golden):
    """`mod_btn` = 'style_bottleneck_IR' / 'styleBlock' (reference src/ops/SAMM/helpers.py:22-57): the extractors alone (16 -> 16 and
    16 -> 32 channels: identity and 1x1 shortcut) and inside StyledscaleNshfitBlock in front of the alignment, state dicts loaded strictly
    from the reference modules' own (tests/golden/make_golden.py modbtn)."""
    from oodgan import samm
    g = golden('modbtn.npz')
    x, style, gen = g['x'].to(dev), g['style'].to(dev), g['gen'].to(dev)

    def state(tag):
        pre = tag + '.sd.'
        return {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}

    for tag, mod in (('sb', samm.style_bottleneck_IR(16, 16, 32, bn=False)), ('sb2', samm.style_bottleneck_IR(16, 32, 32, bn=False)),
                     ('blk', samm.styleBlock(16, 16, 32, noiseInjection=False, activation=False))):
        mod.load_state_dict(state(tag), strict=True)
        close(mod.to(dev)(x, style), g[tag + '.y'], 2e-4)
    for tag, btn in (('blockA', 'style_bottleneck_IR'), ('blockB', 'styleBlock')):
        blk = samm.StyledscaleNshfitBlock(16, 16, 32, btn=btn, scale=0.08, cycle_align=1, diff_fAndg=True)
        blk.load_state_dict(state(tag), strict=True)
        y, f = blk.to(dev)(x, style, image=gen, aligned=None)
        close(y, g[tag + '.y'], 3e-4)
        close(f, g[tag + '.field'], 3e-4)


@pytest.mark.parametrize('wscale', [1.0, 1e-3, 3e5])
def test_bottleneck_8wave_path_range_and_batch(dev, wscale):
    """The AlignNet bottleneck on the S-form + 8-wave kernels (>= 64 channels), batch 3, against the oracle — with the first
    conv's weights scaled so that the UN-normalised PReLU(conv) handed to the second S-form conversion is ~1e-3 or ~3e5 (beyond the f16 maximum): the
    fp32 reference has no range limit (the InstanceNorm behind it undoes the scale), an unscaled f16 pair would lose its lo
    half / overflow."""
    from oodgan import samm
    from oodgan.synth import _bottleneck
    from collections import OrderedDict
    C, B, H = 64, 3, 32
    sd = OrderedDict()
    _bottleneck(sd, 'b', C, C, 77)
    sd['b.res_layer.1.weight'] = sd['b.res_layer.1.weight'] * wscale
    x = synth.normal('bn.x', (B, C, H, H), 78, 1.5, 0.2)
    ref = R.bottleneck_ir(sd, 'b', x)
    m = samm.bottleneck_IR(C, C, 1, 'InstanceNorm', False)
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(dev)
    y = m(x.to(dev))
    close(y, ref, 2e-5)
    y1 = m(x[1:2].contiguous().to(dev))                   # sample 1 alone == sample 1 in the batch
    close(y1, ref[1:2], 2e-5)


def test_mask_blend_vs_oracle(dev):
    from oodgan import samm
    B = 2
    aligns = {k + 1: torch.cat([synth.normal(f'mb.d{k}', (B, 2, s, s), 1, 0.05), synth.uniform(f'mb.a{k}', (B, 1, s, s), 2)], 1)
              for k, s in enumerate((32, 64, 128, 256))}
    x = synth.make_images(1024, B, seed=3)
    gen = synth.normal('mb.gen', (B, 3, 1024, 1024), 4)
    alpha_ref = R.blending_mask(aligns, 1024)
    out_ref = alpha_ref * x + gen * (1 - alpha_ref)
    alpha, out = samm.mask_blend([aligns[k].to(dev) for k in sorted(aligns)], x.to(dev), gen.to(dev), 1024)
    close(alpha, alpha_ref, 1e-5)
    close(out, out_ref, 1e-5)
    strip = samm.extract_masks({k: v.to(dev) for k, v in aligns.items()})
    assert torch.equal(strip.cpu(), R.extract_masks(aligns))


def test_ood_forward_1024_vs_golden(dev, golden):
    """The whole path after the encoder at 1024², B=1, against vectors produced by the reference."""
    from oodgan.arch import ood_faceGAN_e4e
    g = golden('ood_1024.npz')
    m = ood_faceGAN_e4e(out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08,
                        cycle_align=2, blend_with_gen=True, ModSize=256, build_encoder=False)
    res = m.load_state_dict(synth.ood_state(1024, seed=31), strict=True)
    m = m.to(dev).eval()
    enc_lats = synth.make_latents(1024, 1, seed=32, std=0.3).to(dev)
    enc_feats = [f.to(dev) for f in synth.make_encoder_feats(1, seed=33)]
    x = synth.make_images(1024, 1, seed=34).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(1024, 1, seed=35)]
    out, lats = m(x, enc_lats=enc_lats, enc_feats=enc_feats, noise=noises)
    # north_star bar: |Δpixel| < 1e-3 ABSOLUTE on the output (max |out| here is ~3)
    def absclose(a, b, tol, what):
        a = a.detach().cpu()
        assert a.shape == b.shape, (what, a.shape, b.shape)
        err = (a - b).abs().max().item()
        print(f'ood_1024 {what}: max abs err {err:.3e} (ref absmax {b.abs().max().item():.3f})')
        assert err < tol, (what, err)

    absclose(lats, g['lats'], 1e-6, 'lats')
    absclose(out[:, :, ::16, ::16], g['out_sub'], 1e-3, 'out ::16')
    absclose(out[:, :, 480:544, 480:544], g['out_crop'], 1e-3, 'out centre crop')
    absclose(out.double().mean(dim=(2, 3)).float(), g['out_mean'], 1e-5, 'per-channel mean')
    absclose(out.double().std(dim=(2, 3)).float(), g['out_std'], 1e-5, 'per-channel std')
    assert abs(out.abs().max().item() - g['out_absmax'].item()) < 1e-3
    for k in (1, 2, 3, 4):
        a = m.aligns[k]
        step = max(1, a.shape[-1] // 32)
        absclose(a[:, :, ::step, ::step], g[f'align{k}_sub'], 1e-3, f'aligns[{k}]')
        absclose(a.mean(dim=(2, 3)), g[f'align{k}_mean'], 1e-4, f'aligns[{k}] mean')
    absclose(m.aligns[1024][:, :, ::16, ::16], g['align1024_sub'], 1e-3, 'aligns[1024]')
    absclose(m.aligns[1024][:, :1, 480:544, 480:544], g['align1024_crop'], 1e-3, 'aligns[1024] crop')
    from oodgan import samm
    strip = samm.extract_masks(m.aligns)
    absclose(strip[:, :, ::16, ::16], g['mask_strip_sub'], 1e-3, 'mask strip')
    absclose(strip[:, :, 500:502, :], g['mask_strip_rows'], 1e-3, 'mask strip rows')


def test_ood_forward_1024_batch8_image_k_vs_golden(dev, golden):
    """The OOD forward as ``bench.py`` / ``invert`` run it — B=8 — reproduces, for the image in slot 5, the vectors the
    reference produced for that image ALONE (``ood_1024.npz``): SAMM / AlignNet (S-form + 8-wave convs with the PReLU
    epilogue at batch > 1), warp, mask compose and blend are per-sample.  Also through ``invert(steps=0)`` and after two
    W+ steps' worth of stream set-up (``steps=0`` must return the encoder latents untouched)."""
    from oodgan.arch import ood_faceGAN_e4e
    g = golden('ood_1024.npz')
    B, k = 8, 5
    m = ood_faceGAN_e4e(out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08,
                        cycle_align=2, blend_with_gen=True, ModSize=256, build_encoder=False)
    m.load_state_dict(synth.ood_state(1024, seed=31), strict=True)
    m = m.to(dev).eval()

    def batch(make, gold_seed):
        parts = [make(100 * gold_seed + j) for j in range(B)]
        parts[k] = make(gold_seed)
        return parts

    enc_lats = torch.cat(batch(lambda sd: synth.make_latents(1024, 1, seed=sd, std=0.3), 32)).to(dev)
    feats = batch(lambda sd: synth.make_encoder_feats(1, seed=sd), 33)
    enc_feats = [torch.cat([f[i] for f in feats]).to(dev) for i in range(4)]
    x = torch.cat(batch(lambda sd: synth.make_images(1024, 1, seed=sd), 34)).to(dev)
    nz = batch(lambda sd: synth.make_noises(1024, 1, seed=sd), 35)
    noises = [torch.cat([n[i] for n in nz]).to(dev) for i in range(17)]

    def check(out, lats, tag):
        def absclose(a, b, tol, what):
            a = a.detach().cpu()
            assert a.shape == b.shape, (what, a.shape, b.shape)
            err = (a - b).abs().max().item()
            print(f'ood_1024 B=8 slot {k} [{tag}] {what}: max abs err {err:.3e}')
            assert err < tol, (tag, what, err)
        absclose(lats[k:k + 1], g['lats'], 1e-6, 'lats')
        absclose(out[k:k + 1, :, ::16, ::16], g['out_sub'], 1e-4, 'out ::16')
        absclose(out[k:k + 1, :, 480:544, 480:544], g['out_crop'], 1e-4, 'out centre crop')
        absclose(out[k:k + 1].double().mean(dim=(2, 3)).float(), g['out_mean'], 1e-5, 'per-channel mean')
        for lv in (1, 2, 3, 4):
            a = m.aligns[lv]
            assert a.shape[0] == B
            step = max(1, a.shape[-1] // 32)
            absclose(a[k:k + 1, :, ::step, ::step], g[f'align{lv}_sub'], 1e-4, f'aligns[{lv}]')
        absclose(m.aligns[1024][k:k + 1, :, ::16, ::16], g['align1024_sub'], 1e-4, 'aligns[1024]')
        assert torch.isfinite(out).all()

    out, lats = m(x, enc_lats=enc_lats, enc_feats=enc_feats, noise=noises)
    check(out, lats, 'forward')
    prev = None
    for streams in (1, 3):
        out0, lats0, losses = m.invert(x, steps=0, noise=noises, streams=streams, enc_lats=enc_lats, enc_feats=enc_feats)
        assert losses.shape == (0, B) and torch.equal(lats0, lats)
        check(out0, lats0, f'invert(steps=0, streams={streams})')
        # round 4: the first forward of a batch size measures its range scales, the following ones carry them and run the fused
        # producers of the W+ loop (modules.Generator.forward): same values to rounding against the first call, and — same launches on
        # the same stream — bit-identical among themselves
        d = (out0 - out).abs().max().item()
        print(f'carried-scale forward vs measured-scale forward: max abs diff {d:.3e}')
        assert d <= 2e-5
        if prev is not None:
            assert torch.equal(out0, prev)
        prev = out0.clone()
