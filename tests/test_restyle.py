"""ReStyle variant of the OOD forward: ``ood_faceGAN_restyle`` against vectors produced by the
reference's own ``ood_faceGAN_restyle`` (tests/golden/make_golden.py gold_restyle) and host-side checks."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ood-gan-inversion_amd'))

from oodgan import synth  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _close(a, b, tol):
    a = a.detach().float().cpu().numpy()
    b = np.asarray(b.numpy() if torch.is_tensor(b) else b, dtype=np.float32)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float(np.abs(a - b).max())
    assert err <= tol, f'max |diff| {err:.3e} > {tol:.1e}'


def test_restyle_encoder_keys_and_shapes():
    """The mirror has the parameters of the reference encoder (restyle_e4e_encoder.py:37-83): the checkpoint the
    reference loaded with strict=True when the golden was made loads strictly here, and the 18 heads sit on 16x16."""
    from oodgan.encoder import ProgressiveBackboneEncoder
    ck = synth.restyle_checkpoint(seed=51)
    enc = ProgressiveBackboneEncoder(50, 'ir_se', 18, ck['opts'])
    sd = {k[len('encoder.'):]: v for k, v in ck['state_dict'].items()}
    enc.load_state_dict(sd, strict=True)
    assert enc.input_layer[0].weight.shape == (64, 6, 3, 3)
    assert len(enc.styles) == 18 and all(len([m for m in s.convs if isinstance(m, torch.nn.Conv2d)]) == 4 for s in enc.styles)
    assert enc.channels == [64, 64, 128, 256, 512]
    assert tuple(ck['latent_avg'].shape) == (18, 512)


def test_restyle_registry_and_ctor_errors(tmp_path):
    from oodgan.arch import ARCH_REGISTRY, build_network
    assert 'ood_faceGAN_restyle' in ARCH_REGISTRY
    with pytest.raises(AssertionError):
        build_network({'type': 'ood_faceGAN_restyle', 'out_size': 1024})
    pth = tmp_path / 'r.pth'
    torch.save({'state_dict': {}, 'latent_avg': torch.zeros(18, 512), 'opts': {'encoder_type': 'ResNetProgressiveBackboneEncoder', 'input_nc': 6}}, pth)
    with pytest.raises(NotImplementedError):
        build_network({'type': 'ood_faceGAN_restyle', 'out_size': 1024, 'ReStyle_pth': str(pth)})


@pytest.mark.gpu
def test_avgpool_matches_adaptive_avg_pool(dev):
    from oodgan import samm
    x = synth.make_images(64, 2, seed=3)
    y = samm.avgpool(x.to(dev), 16)
    _close(y, torch.nn.functional.adaptive_avg_pool2d(x, (16, 16)).numpy(), 1e-6)


@pytest.mark.gpu
def test_restyle_forward_1024_vs_golden(dev, golden, tmp_path):
    """average image -> 2 encoder cycles with one plain reconstruction between -> OOD forward, B=1, 1024²."""
    from oodgan.arch import build_network
    g = golden('restyle_1024.npz')
    pth = tmp_path / 'restyle.pth'
    torch.save(synth.restyle_checkpoint(seed=51), pth)
    m = build_network(dict(type='ood_faceGAN_restyle', out_size=1024, style_dim=512, encoder='ReStyle', ReStyle_pth=str(pth), enc_cycle=2,
                           enable_modulation=True, warp_scale=0.08, cycle_align=2, blend_with_gen=True, ModSize=256))
    sd = synth.ood_state(1024, seed=31)
    sd.pop('avg_latent')
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.startswith('encoder.') or k == 'avg_latent' for k in res.missing_keys)
    m = m.to(dev).eval()
    x = synth.make_images(1024, 1, seed=52).to(dev)
    passes = [[n.to(dev) for n in synth.make_noises(1024, 1, seed=s)] for s in (53, 54, 55)]
    out, lats = m(x, noise_passes=passes)
    _close(m.avg_img[:, :, ::4, ::4], g['avg_img_sub'], 1e-4)
    _close(lats, g['lats'], 1e-5)
    tol = 2e-4          # measured 2e-5 (lats 7e-7)
    _close(out[:, :, ::16, ::16], g['out_sub'], tol)
    _close(out[:, :, 480:544, 480:544], g['out_crop'], tol)
    for k in (1, 2, 3, 4):
        a = m.aligns[k]
        step = max(1, a.shape[-1] // 32)
        _close(a[:, :, ::step, ::step], g[f'align{k}_sub'], tol)
    _close(m.aligns[1024][:, :1, ::16, ::16], g['align1024_sub'], tol)
    # second call reuses the cached average image; same noise for the remaining passes -> same result.  Round 4: the generator passes of
    # a later call run on carried range scales and the W+ loop's fused producers (modules.Generator.forward), the very first one of a
    # batch size on measured scales: equal to fp32 rounding against the first call, bit-identical from then on
    out2, lats2 = m(x, noise_passes=passes)
    lats2 = lats2.clone()
    assert (lats2 - lats).abs().max().item() <= 1e-5
    out3, lats3 = m(x, noise_passes=passes)
    assert torch.equal(lats3, lats2)


# ------------------------------------------------------------------ Feature-Style variant
def test_fs_encoder_keys_and_cpu_forward_shapes():
    """Mirror of fs_encoder_v2 (feature_style_encoder.py:12-74): the state the reference loaded strictly loads strictly
    here; output contract (latents, content, 4 SAMM taps) on a small input."""
    from oodgan.encoder import fs_encoder_v2
    enc = fs_encoder_v2(18, stride=(2, 2)).eval()
    enc.load_state_dict(synth.featurestyle_state(seed=61), strict=True)
    import torch_encoder_mirror as TM
    with torch.no_grad():
        lats, content, taps = TM.fs_encoder_forward(enc, synth.make_images(64, 1, seed=5), return_feats=True)
    assert lats.shape == (1, 18, 512) and content.shape == (1, 512, 4, 4)
    assert [tuple(t.shape[1:]) for t in taps] == [(64, 64, 64), (64, 32, 32), (128, 16, 16), (256, 8, 8)]
    assert enc.styles[0].weight.shape == (512, 960 * 9)


@pytest.mark.gpu
def test_adaptive_avgpool_3x3(dev):
    from oodgan import samm
    for s in (128, 64, 32, 16, 7):
        x = synth.normal(f'ap.{s}', (2, 5, s, s), 3)
        _close(samm.avgpool(x.to(dev), 3), torch.nn.functional.adaptive_avg_pool2d(x, (3, 3)), 2e-6)


@pytest.mark.gpu
def test_featurestyle_forward_1024_vs_golden(dev, golden, tmp_path):
    from oodgan.arch import build_network
    g = golden('featurestyle_1024.npz')
    pth, avg = tmp_path / 'fs.pth', tmp_path / 'avg.pth'
    torch.save(synth.featurestyle_state(seed=61), pth)
    torch.save(synth.normal('fs.latent_avg', (18, 512), 61, 0.5), avg)
    m = build_network(dict(type='ood_faceGAN_FeatureStyle', out_size=1024, style_dim=512, encoder='FeatureStyle', FeatureStyle_pth=str(pth),
                           arcface_model_path=None, avg_latent_pth=str(avg), enable_modulation=True, warp_scale=0.08, cycle_align=2,
                           blend_with_gen=True, ModSize=256))
    sd = synth.ood_state(1024, seed=31)
    sd.pop('avg_latent')
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.startswith('encoder.') or k == 'avg_latent' for k in res.missing_keys)
    m = m.to(dev).eval()
    x = synth.make_images(1024, 1, seed=62).to(dev)
    noise = [n.to(dev) for n in synth.make_noises(1024, 1, seed=63)]
    out, lats = m(x, noise=noise)
    _, content, taps = m.encoder(m.face_pool(x), return_feats=True)
    for i, f in enumerate(taps):
        _close(f.mean(dim=(2, 3)), g[f'tap{i}_mean'], 2e-4)
    _close(content[:, ::8], g['content_sub'], 1e-3)
    _close(lats, g['lats'], 2e-4)
    tol = 1e-3
    _close(out[:, :, ::16, ::16], g['out_sub'], tol)
    _close(out[:, :, 480:544, 480:544], g['out_crop'], tol)
    for k in (1, 2, 3, 4):
        a = m.aligns[k]
        step = max(1, a.shape[-1] // 32)
        _close(a[:, :, ::step, ::step], g[f'align{k}_sub'], tol)
    _close(m.aligns[1024][:, :1, ::16, ::16], g['align1024_sub'], tol)
