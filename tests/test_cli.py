"""The harness around the path: YAML option surface of run_ood_faceGAN_inversion.py, checkpoint
filter of load_model, per-image outputs (inversion PNG + mask strip) and metrics."""
import os

import numpy as np
import pytest
import torch
import yaml

from oodgan import synth


def _options(tmp, with_ckpt=True):
    return {
        'name': 'OOD_faceGAN_e4e', 'save_dir': str(tmp / 'results'), 'directions_dir': str(tmp / 'directions'),
        'datasets': {'val_0': {'dataroot': str(tmp / 'data'), 'editing': {'direction': 'Smiling', 'intensity': 2}},
                     'val_1': {'dataroot': str(tmp / 'data')}},
        'network_g': {'type': 'ood_faceGAN_e4e', 'out_size': 1024, 'style_dim': 512, 'encoder': 'E4E', 'enable_modulation': True,
                      'warp_scale': 0.08, 'cycle_align': 2, 'blend_with_gen': True, 'ModSize': 256},
        'path': {'pretrain_network_g': str(tmp / 'net_g.pth') if with_ckpt else None, 'param_key_g': 'params_ema', 'strict_load_g': False},
        'metrics': {'lpips': {'crop_border': 2, 'test_y_channel': False}, 'psnr': {'crop_border': 2, 'test_y_channel': False},
                    'ssim': {'crop_border': 2, 'test_y_channel': False}},
    }


def test_option_surface_and_registry(tmp_path):
    from oodgan import cli
    opts = _options(tmp_path, with_ckpt=False)
    text = yaml.safe_dump(opts)
    assert yaml.load(text, Loader=yaml.FullLoader) == opts
    assert set(cli.model_dict) == {'ood_faceGAN_e4e', 'ood_faceGAN_restyle', 'ood_faceGAN_FeatureStyle'}   # run_ood_faceGAN_inversion.py:23-27
    bad = dict(opts, network_g=dict(opts['network_g'], type='ood_faceGAN_pSp'))
    with pytest.raises(KeyError):
        cli.load_model(bad)
    os.makedirs(tmp_path / 'data')
    for n in ('b.png', 'a.jpg', 'c.txt'):
        (tmp_path / 'data' / n).write_bytes(b'')
    files, direction = cli.load_files_from_path({'dataroot': str(tmp_path / 'data')})
    assert [os.path.basename(f) for f in files] == ['a.jpg', 'b.png'] and direction.item() == 0.0


# The ``network_g`` / ``path`` blocks of the three YAMLs the reference ships under options/test/ (E4E_Face_test.yml,
# ReStyle_Face_test.yml, FeatureStyle_Face_test.yml), option for option; only the checkpoint locations are replaced by
# {placeholders} that the test fills with recipe checkpoints of the same formats.
SHIPPED_YAML = {
    'ood_faceGAN_e4e': '''
name: OOD_faceGAN_e4e
save_dir: ./results
directions_dir: ./directions
network_g:
  type: ood_faceGAN_e4e
  out_size: 1024
  style_dim: 512
  StyleGAN_pth: {stylegan}
  StyleGAN_pth_key: g_ema
  avg_latent_pth: {avg1}
  E4E_pth: {e4e}
  encoder: E4E
  enable_modulation: true
  warp_scale: 0.08
  cycle_align: 2
  blend_with_gen: true
  ModSize: 256
path:
  pretrain_network_g: {net_g}
  param_key_g: 'params_ema'
  strict_load_g: false
''',
    'ood_faceGAN_restyle': '''
name: OOD_faceGAN_ReStyle
save_dir: ./results
directions_dir: ./directions
network_g:
  type: ood_faceGAN_restyle
  out_size: 1024
  style_dim: 512
  StyleGAN_pth: {stylegan}
  StyleGAN_pth_key: g_ema
  avg_latent_pth: {avg1}
  ReStyle_pth: {restyle}
  encoder: ReStyle
  enc_cycle: 5
  enable_modulation: true
  warp_scale: 0.08
  cycle_align: 2
  blend_with_gen: true
  ModSize: 256
path:
  pretrain_network_g: {net_g}
  param_key_g: 'params_ema'
  strict_load_g: false
''',
    'ood_faceGAN_FeatureStyle': '''
name: OOD_faceGAN_FeatureStyle
save_dir: ./results
directions_dir: ./directions
network_g:
  type: ood_faceGAN_FeatureStyle
  out_size: 1024
  style_dim: 512
  StyleGAN_pth: {stylegan}
  StyleGAN_pth_key: g_ema
  avg_latent_pth: {avg18}
  FeatureStyle_pth: {fs}
  arcface_model_path: {arc}
  enable_modulation: true
  warp_scale: 0.08
  cycle_align: 2
  blend_with_gen: true
  ModSize: 256
path:
  pretrain_network_g: {net_g}
  param_key_g: 'params_ema'
  strict_load_g: false
''',
}


@pytest.fixture(scope='module')
def shipped_ckpts(tmp_path_factory):
    """Recipe checkpoints in the formats the three YAMLs point at."""
    from oodgan.arch import ood_faceGAN_e4e
    d = tmp_path_factory.mktemp('ckpt')
    gen = synth.generator_state(1024, seed=5)
    torch.save({'g_ema': gen}, d / 'stylegan.pth')
    torch.save(synth.normal('avg1', (1, 512), 3, 0.5), d / 'avg1.pth')
    torch.save(synth.normal('avg18', (18, 512), 3, 0.5), d / 'avg18.pth')
    m = ood_faceGAN_e4e(out_size=1024)
    enc = synth.encoder_state({k: tuple(v.shape) for k, v in m.encoder.state_dict().items()}, seed=41)
    torch.save({'state_dict': {'encoder.' + k: v for k, v in enc.items()}}, d / 'e4e.pt')
    torch.save(synth.restyle_checkpoint(seed=51), d / 'restyle.pt')
    torch.save(synth.featurestyle_state(seed=61), d / 'fs.pth')
    # the SAMM checkpoint of load_model: modulation.* / feats_conv.* only, plus a stale 2-D delta_latent it must drop
    samm_sd = {k: v for k, v in synth.ood_state(1024, seed=31).items() if k.startswith(('modulation.', 'feats_conv.'))}
    samm_sd['delta_latent'] = torch.ones(18, 512)
    torch.save({'params_ema': samm_sd}, d / 'net_g.pth')
    return dict(stylegan=d / 'stylegan.pth', avg1=d / 'avg1.pth', avg18=d / 'avg18.pth', e4e=d / 'e4e.pt', restyle=d / 'restyle.pt',
                fs=d / 'fs.pth', arc=d / 'absent_backbone.pth', net_g=d / 'net_g.pth'), gen, samm_sd


@pytest.mark.parametrize('variant', sorted(SHIPPED_YAML))
def test_shipped_yaml_variants_resolve(variant, shipped_ckpts):
    """run_ood_faceGAN_inversion.py:23-47 for each YAML the reference ships: ``type`` resolves through ``model_dict``, the
    remaining ``network_g`` keys are the constructor's kwargs, the component checkpoints load, ``load_model`` applies the
    SAMM checkpoint non-strictly and zeroes ``delta_latent``."""
    from oodgan import arch, cli
    paths, gen, samm_sd = shipped_ckpts
    opts = yaml.load(SHIPPED_YAML[variant].format(**paths), Loader=yaml.FullLoader)
    assert opts['network_g']['type'] == variant
    m = cli.load_model(opts)
    assert type(m) is getattr(arch, variant) and type(m) is arch.ARCH_REGISTRY.get(variant)
    assert opts['network_g']['type'] == variant                      # the caller's dict is not consumed
    assert m.generator.size == 1024 and m.ModSize == 256 and m.warp_scale == 0.08 and m.cycle_align == 2 and m.blend_with_gen
    assert len(m.modulation) == 4 and len(m.feats_conv) == 4
    assert torch.equal(m.generator.state_dict()['convs.15.conv.weight'], gen['convs.15.conv.weight'])
    k = 'modulation.0.alignment.align_net.body.0.res_layer.1.weight'
    k = k if k in samm_sd else next(n for n in samm_sd if n.startswith('modulation.0.') and samm_sd[n].dim() == 4)
    assert torch.equal(m.state_dict()[k], samm_sd[k])
    assert m.delta_latent.shape == (1, 18, 512) and not m.delta_latent.any()
    if variant == 'ood_faceGAN_restyle':
        assert m.enc_cycle == 5 and m.avg_latent.shape == (18, 512) and m.encoder_type == 'ReStyle'
    elif variant == 'ood_faceGAN_FeatureStyle':
        assert m.avg_latent.shape == (18, 512) and m.encoder_type == 'FeatureStyle'
    else:
        assert m.avg_latent.shape == (1, 512) and m.encoder_type == 'E4E'
        assert torch.equal(m.avg_latent.data, torch.load(paths['avg1']))


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path):
    from oodgan import cli, imgio
    from oodgan.arch import ood_faceGAN_e4e
    opts = _options(tmp_path)
    # checkpoint: recipe weights for generator + SAMM + encoder, plus a stale 2-D delta_latent that load_model must drop
    m = ood_faceGAN_e4e(**{k: v for k, v in opts['network_g'].items() if k != 'type'})
    sd = synth.ood_state(1024, seed=31)
    enc = synth.encoder_state({k: tuple(v.shape) for k, v in m.encoder.state_dict().items()}, seed=41)
    sd.update({'encoder.' + k: (v * 0.1 if k.endswith('linear.weight') else v) for k, v in enc.items()})
    sd['delta_latent'] = torch.ones(18, 512)
    torch.save({'params_ema': sd}, tmp_path / 'net_g.pth')
    os.makedirs(tmp_path / 'directions')
    np.save(tmp_path / 'directions' / 'Smiling.npy', (0.01 * np.random.default_rng(0).standard_normal((18, 512))).astype(np.float32))
    os.makedirs(tmp_path / 'data')
    rng = np.random.default_rng(1)
    for n in ('00002.png', '00001.png'):
        # BASELINE configs[0]: a single 256x256 face through the E4E_Face_test.yml option surface
        imgio.imwrite(str(tmp_path / 'data' / n), rng.integers(0, 256, (256, 256, 3), dtype=np.uint8))
    with open(tmp_path / 'opt.yml', 'w') as f:
        yaml.safe_dump(opts, f)
    summary = cli.main(['--opt', str(tmp_path / 'opt.yml'), '--wplus-steps', '2'])
    assert set(summary) == {'val_0', 'val_1'}
    for name in summary:
        s = summary[name]
        assert s['n'] == 2 and np.isfinite(s['psnr']) and 0 < s['ssim'] <= 1 and s['time'] > 0
        for n in ('00001.png', '00002.png'):
            inv = imgio.imread(str(tmp_path / 'results' / 'OOD_faceGAN_e4e' / name / 'inversion' / n))
            assert inv.shape == (1024, 1024, 3)
            from PIL import Image
            with Image.open(tmp_path / 'results' / 'OOD_faceGAN_e4e' / name / 'masks' / n) as im:
                assert im.size == (5 * 1024, 1024)                # levels 1..4 and the composed 1024 mask, side by side
    # round 6: `inversion.batch` — both files in ONE call of the model.  (Pixels are not comparable with the per-file run: without `noise=` the
    # generator draws fresh noise maps per call, as the reference does — model.py:505-508 — so only the files, their geometry and the metrics'
    # neighbourhood are checked; that a batched inversion equals the per-image one is tests/test_hip_wplus_long.py's subject.)
    opts['inversion'] = dict(opts.get('inversion') or {}, batch=2)
    opts['save_dir'] = str(tmp_path / 'results_b2')
    with open(tmp_path / 'opt_b2.yml', 'w') as f:
        yaml.safe_dump(opts, f)
    summary2 = cli.main(['--opt', str(tmp_path / 'opt_b2.yml'), '--wplus-steps', '2'])
    for name in summary2:
        assert summary2[name]['n'] == 2 and abs(summary2[name]['psnr'] - summary[name]['psnr']) < 1.0 and summary2[name]['time'] > 0
        for n in ('00001.png', '00002.png'):
            a = imgio.imread(str(tmp_path / 'results_b2' / 'OOD_faceGAN_e4e' / name / 'inversion' / n))
            assert a.shape == (1024, 1024, 3)
            with Image.open(tmp_path / 'results_b2' / 'OOD_faceGAN_e4e' / name / 'masks' / n) as im:
                assert im.size == (5 * 1024, 1024)
        # the two images of a batch are different inversions (their own mask strips), not copies of item 0
        m1 = imgio.imread(str(tmp_path / 'results_b2' / 'OOD_faceGAN_e4e' / name / 'masks' / '00001.png')).astype(np.int32)
        m2 = imgio.imread(str(tmp_path / 'results_b2' / 'OOD_faceGAN_e4e' / name / 'masks' / '00002.png')).astype(np.int32)
        assert np.abs(m1 - m2).max() > 0


@pytest.mark.gpu
def test_cli_c1_uint8_outputs_vs_reference(tmp_path, golden, monkeypatch):
    """BASELINE configs[0] at the pixel level: the PNGs the CLI writes for one 256x256 image — inversion
    and mask strip — against the uint8 arrays the reference pipeline produces for the same file (tests/golden/make_golden.py
    gold_cli_c1: real img2tensor / F.interpolate / ood_faceGAN_e4e incl. encoder / tensor2img with its ``.round()``).
    The only way to differ is a float within rounding of a .5 boundary: at most 1 LSB, on at most 5e-4 of the values
    (measured: 37 of 196 608 = 1.9e-4 in the image, 0-4 values per mask level)."""
    from oodgan import cli, imgio, modules
    from PIL import Image
    g = {k: v.numpy() for k, v in golden('cli_c1.npz').items()}
    opts = _options(tmp_path)
    opts['datasets'] = {'val_1': {'dataroot': str(tmp_path / 'data')}}
    m = cli.model_dict['ood_faceGAN_e4e'](**{k: v for k, v in opts['network_g'].items() if k != 'type'})
    sd = synth.ood_state(1024, seed=31)
    enc = synth.encoder_state({k: tuple(v.shape) for k, v in m.encoder.state_dict().items()}, seed=41)
    sd.update({'encoder.' + k: (v * 0.1 if k.endswith('linear.weight') else v) for k, v in enc.items()})
    torch.save({'params_ema': sd}, tmp_path / 'net_g.pth')
    del m
    imgio.imwrite(str(tmp_path / 'data' / 'face.png'), g['bgr'])
    assert np.array_equal(imgio.imread(str(tmp_path / 'data' / 'face.png')), g['bgr'])
    with open(tmp_path / 'opt.yml', 'w') as f:
        yaml.safe_dump(opts, f)
    # the reference draws its noise maps from the RNG; the golden run fed it preset maps, the same ones go in here
    presets = synth.make_noises(1024, 1, seed=35)
    draw = modules.Generator._draw_noises
    monkeypatch.setattr(modules.Generator, '_draw_noises', lambda self, batch, noise, rnd: draw(
        self, batch, [presets[i].to(self.input.input.device) if (noise is None or noise[i] is None) else noise[i] for i in range(17)], rnd))
    summary = cli.main(['--opt', str(tmp_path / 'opt.yml')])
    root = tmp_path / 'results' / 'OOD_faceGAN_e4e' / 'val_1'
    res = imgio.imread(str(root / 'inversion' / 'face.png'))
    assert res.shape == (1024, 1024, 3) and res.dtype == np.uint8

    def lsb_check(a, b, what, frac):
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        print(f'c1 {what}: {int((d > 0).sum())} of {d.size} values differ, max {int(d.max())} LSB')
        assert d.max() <= 1 and (d > 0).mean() <= frac, (what, int(d.max()), float((d > 0).mean()))

    lsb_check(res[::4, ::4], g['out_u8_sub'], 'inversion ::4', 5e-4)
    lsb_check(res[448:576, 448:576], g['out_u8_crop'], 'inversion crop', 5e-4)
    with Image.open(root / 'masks' / 'face.png') as im:
        strip = np.asarray(im)
    assert strip.shape == (1024, 5 * 1024) and strip.dtype == np.uint8
    for i, s_ in enumerate((32, 64, 128, 256)):
        st = 1024 // s_
        part = strip[:, 1024 * i:1024 * (i + 1)]
        native = part[::st, ::st]
        assert np.array_equal(np.repeat(np.repeat(native, st, 0), st, 1), part)       # nearest indexing: bit-exact structure
        lsb_check(native, g[f'mask{i + 1}_u8'], f'mask level {i + 1}', 5e-4)
    lsb_check(strip[::4, 4096::4], g['mask1024_u8_sub'], 'mask 1024 ::4', 5e-4)
    lsb_check(strip[448:576, 4096 + 448:4096 + 576], g['mask1024_u8_crop'], 'mask 1024 crop', 5e-4)
    assert abs(summary['val_1']['psnr'] - float(g['psnr_resized_gt'])) < 1e-3
