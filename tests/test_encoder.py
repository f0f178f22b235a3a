"""e4e encoder against vectors produced by the reference encoder (tests/golden/make_golden.py
gold_encoder).  CPU test = the parameter container's structure / key parity, checked numerically through the plain-torch
restatement that lives in tests/ (torch_encoder_mirror.py); GPU tests = ``Encoder4EditingHIP`` (the product's forward) and
the end-to-end ``ood_faceGAN_e4e.forward(x)`` from an image."""
import pytest
import torch

from oodgan import synth
import torch_encoder_mirror as TM


def _build():
    from oodgan.encoder import Encoder4Editing
    enc = Encoder4Editing(50, 'ir_se', {'stylegan_size': 1024}, bn=True).eval()
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert len(shapes) == 621                       # same state-dict size as the reference (probe in SURVEY App. C)
    enc.load_state_dict(synth.encoder_state(shapes, seed=41), strict=True)
    return enc


def _check(w, feats, g, tol):
    def close(a, b):
        a = a.detach().cpu()
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())
    close(w, g['w'])
    assert len(feats) == 5
    for i, f in enumerate(feats):
        step = max(1, f.shape[-1] // 16)
        close(f[:, ::8, ::step, ::step], g[f'feat{i}_sub'])
        close(f.mean(dim=(2, 3)), g[f'feat{i}_mean'])


def test_encoder_cpu_vs_reference_golden(golden):
    g = golden('encoder_256.npz')
    enc = _build()
    x = synth.make_images(256, 1, seed=42)
    with torch.no_grad():
        w, feats = TM.encoder4editing_forward(enc, x, return_feats=True)
    assert enc.channels == [64, 64, 128, 256, 512] and w.shape == (1, 18, 512)
    _check(w, feats, g, 1e-4)
    with pytest.raises(RuntimeError):          # the container itself has no compute path
        enc(x)


@pytest.mark.gpu
def test_arch_forward_from_image_end_to_end():
    """forward(x) with the encoder attached == forward with its outputs passed explicitly; invert() runs."""
    from oodgan.arch import ood_faceGAN_e4e
    dev = torch.device('cuda:0')
    m = ood_faceGAN_e4e(out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08,
                        cycle_align=2, blend_with_gen=True, ModSize=256)
    sd = synth.ood_state(1024, seed=31)
    shapes = {k: tuple(v.shape) for k, v in m.encoder.state_dict().items()}
    enc_sd = synth.encoder_state(shapes, seed=41)
    for k in enc_sd:                      # keep the encoder latents at the scale of trained W+ codes
        if k.endswith('linear.weight'):
            enc_sd[k] = enc_sd[k] * 0.1
    sd.update({'encoder.' + k: v for k, v in enc_sd.items()})
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    x = synth.make_images(1024, 1, seed=34).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(1024, 1, seed=35)]
    out, lats = m(x, noise=noises)
    assert out.shape == (1, 3, 1024, 1024) and lats.shape == (1, 18, 512) and torch.isfinite(out).all()
    with torch.no_grad():
        from oodgan import samm
        el, ef = m.encoder(samm.resize_bilinear(x, 256), return_feats=True)
    out2, lats2 = m(x, noise=noises, enc_lats=el, enc_feats=ef)
    # MIOpen may pick a different algorithm on the second encoder call: same values up to fp32 rounding, not bit-equal
    assert (lats - lats2).abs().max().item() <= 1e-4 * max(1.0, lats.abs().max().item())
    assert (out - out2).abs().max().item() < 1e-3
    assert sorted(m.aligns.keys()) == [1, 2, 3, 4, 1024]
    out3, lats3, losses = m.invert(x, steps=3, noise=noises)
    assert losses.shape == (3, 1) and losses[-1].item() < losses[0].item() and torch.isfinite(out3).all()


@pytest.mark.gpu
def test_hip_encoder_vs_reference_golden_and_torch_mirror(golden):
    """The encoder on the HIP kernels (oodgan/encoder_hip.py): same state dict, same outputs as the reference."""
    from oodgan.encoder_hip import Encoder4EditingHIP
    g = golden('encoder_256.npz')
    dev = torch.device('cuda:0')
    ref_mod = _build()
    enc = Encoder4EditingHIP(50, 'ir_se', {'stylegan_size': 1024}, bn=True).eval()
    assert list(enc.state_dict().keys()) == list(ref_mod.state_dict().keys())
    enc.load_state_dict(ref_mod.state_dict(), strict=True)
    enc = enc.to(dev)
    x = synth.make_images(256, 1, seed=42).to(dev)
    w, feats = enc(x, return_feats=True)
    _check(w, feats, g, 2e-4)
    # batch of 2 (second image different): rows independent
    x2 = torch.cat([x, synth.make_images(256, 1, seed=43).to(dev)])
    w2 = enc(x2)
    assert w2.shape == (2, 18, 512) and (w2[:1] - w).abs().max().item() <= 1e-4 * w.abs().max().item()


@pytest.mark.gpu
def test_graphed_forward_equals_eager():
    """oodgan.arch.GraphedForward: model(x) replayed from a captured hipGraph gives the eager result bit for bit (fixed noise maps, same
    range-scale history), follows new inputs, and refreshes model.aligns."""
    from oodgan.arch import GraphedForward, ood_faceGAN_e4e
    dev = torch.device('cuda:0')
    m = ood_faceGAN_e4e(out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08,
                        cycle_align=2, blend_with_gen=True, ModSize=256)
    sd = synth.ood_state(1024, seed=31)
    enc_sd = synth.encoder_state({k: tuple(v.shape) for k, v in m.encoder.state_dict().items()}, seed=41)
    sd.update({'encoder.' + k: (v * 0.1 if k.endswith('linear.weight') else v) for k, v in enc_sd.items()})
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).eval()
    noises = [n.to(dev) for n in synth.make_noises(1024, 1, seed=35)]
    gf = GraphedForward(m)
    for seed in (34, 36):
        x = synth.make_images(1024, 1, seed=seed).to(dev)
        # round 4: a forward carries the generator's range scales from the previous one (the first of a batch size measures them), so the
        # bits depend on what ran before: the reference is the second of two eager calls — scales of this very input, as the replay has
        first, _ = m(x, noise=noises)
        first = first.clone()
        ref, ref_lats = m(x, noise=noises)
        assert (ref - first).abs().max().item() <= 2e-5
        ref, ref_lats, ref_mask = ref.clone(), ref_lats.clone(), m.aligns[1024].clone()
        out, lats = gf(x, noise=noises)
        assert torch.equal(out, ref) and torch.equal(lats, ref_lats) and torch.equal(m.aligns[1024], ref_mask)
    assert len(gf._cache) == 1
