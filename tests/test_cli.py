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
    bad = dict(opts, network_g=dict(opts['network_g'], type='ood_faceGAN_restyle'))
    with pytest.raises(KeyError):
        cli.load_model(bad)
    os.makedirs(tmp_path / 'data')
    for n in ('b.png', 'a.jpg', 'c.txt'):
        (tmp_path / 'data' / n).write_bytes(b'')
    files, direction = cli.load_files_from_path({'dataroot': str(tmp_path / 'data')})
    assert [os.path.basename(f) for f in files] == ['a.jpg', 'b.png'] and direction.item() == 0.0


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
