"""Checkpoint formats of the constructor path (reference OOD_faceGAN_e4e_arch.py:137-153): the generator
file in rosinality layout (``StyleGAN_pth_key='g_ema'``) and in BasicSR layout (``'params_ema'``, keys as produced by
BasicSR/scripts/model_conversion/convert_stylegan.py), the e4e ``.pt`` (``state_dict`` with ``encoder.*`` keys among
others), ``avg_latent_pth`` and ``delta_latent_pth`` tensors — loaded through ``ood_faceGAN_e4e(StyleGAN_pth=..., E4E_pth=...,
avg_latent_pth=..., delta_latent_pth=...)`` and compared with the same model filled by a strict ``load_state_dict``."""
from collections import OrderedDict

import pytest
import torch

from oodgan import synth


_CACHE = {}


def _write_generator(tmp_path, layout, gen):
    from oodgan.modules import StyleGAN2Generator
    if layout == 'basicsr':
        to_b = StyleGAN2Generator(1024)._ros_to_basicsr
        gen_file = {'params_ema': OrderedDict((to_b(k), v) for k, v in gen.items() if not k.endswith('.kernel'))}
        key = 'params_ema'
    else:
        gen_file = {'g_ema': gen, 'g': OrderedDict((k, torch.zeros_like(v)) for k, v in gen.items())}   # 'g' must be ignored
        key = 'g_ema'
    path = str(tmp_path / f'stylegan2_{layout}.pth')
    torch.save(gen_file, path)
    return path, key


def _recipe(tmp_path_factory, layout):
    if layout in _CACHE:
        return _CACHE[layout]
    if 'dir' not in _CACHE:
        _CACHE['dir'] = tmp_path_factory.mktemp('ckpt')
    tmp_path = _CACHE['dir']
    if 'ref' in _CACHE:         # second layout: only the generator file differs
        ref, sd, kw0 = _CACHE['ref']
        gen = OrderedDict((k[len('generator.'):], v) for k, v in sd.items() if k.startswith('generator.'))
        kw = dict(kw0)
        kw['StyleGAN_pth'], kw['StyleGAN_pth_key'] = _write_generator(tmp_path, layout, gen)
        _CACHE[layout] = (ref, sd, kw, tmp_path)
        return _CACHE[layout]
    from oodgan.arch import ood_faceGAN_e4e
    ref = ood_faceGAN_e4e(out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08, cycle_align=2,
                          blend_with_gen=True, ModSize=256)
    sd = synth.ood_state(1024, seed=31)
    enc = synth.encoder_state({k: tuple(v.shape) for k, v in ref.encoder.state_dict().items()}, seed=41)
    sd.update({'encoder.' + k: v for k, v in enc.items()})
    ref.load_state_dict(sd, strict=True)
    gen = OrderedDict((k[len('generator.'):], v) for k, v in sd.items() if k.startswith('generator.'))
    gpath, key = _write_generator(tmp_path, layout, gen)
    # an e4e checkpoint as pSp writes it: encoder.* next to decoder.* and latent_avg, plus opts
    e4e = OrderedDict(('encoder.' + k, v) for k, v in enc.items())
    e4e['decoder.conv1.conv.weight'] = torch.zeros(1)
    torch.save({'state_dict': e4e, 'latent_avg': torch.zeros(18, 512), 'opts': {'stylegan_size': 1024}}, tmp_path / 'e4e.pt')
    torch.save(sd['avg_latent'].clone(), tmp_path / 'avg.pth')
    torch.save(sd['delta_latent'].clone(), tmp_path / 'delta.pth')
    kw = dict(StyleGAN_pth=gpath, StyleGAN_pth_key=key, E4E_pth=str(tmp_path / 'e4e.pt'),
              avg_latent_pth=str(tmp_path / 'avg.pth'), delta_latent_pth=str(tmp_path / 'delta.pth'))
    _CACHE['ref'] = (ref, sd, kw)
    _CACHE[layout] = (ref, sd, kw, tmp_path)
    return _CACHE[layout]


@pytest.mark.parametrize('layout', ['rosinality', 'basicsr'])
def test_constructor_checkpoint_paths_fill_the_same_state(tmp_path_factory, layout):
    from oodgan.arch import build_network
    ref, sd, kw, _ = _recipe(tmp_path_factory, layout)
    m = build_network(dict(type='ood_faceGAN_e4e', out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08,
                           cycle_align=2, blend_with_gen=True, ModSize=256, **kw))
    a, b = m.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys())
    filled = [k for k in a if k.startswith(('generator.', 'encoder.')) or k in ('avg_latent', 'delta_latent')]
    assert len(filled) == 621 + 2 + len([k for k in a if k.startswith("generator.")]) and len(filled) > 780
    for k in filled:
        assert torch.equal(a[k], b[k]), k
    # the SAMM weights are NOT in these files (they come with load_network_g, tested in test_cli.py): still at init
    assert not torch.equal(a['modulation.0.alignment.body.body.0.res_layer.1.weight'], b['modulation.0.alignment.body.body.0.res_layer.1.weight'])


def test_missing_key_and_bad_file_are_loud(tmp_path_factory):
    from oodgan.arch import ood_faceGAN_e4e
    ref, sd, kw, tmp_path = _recipe(tmp_path_factory, 'rosinality')
    with pytest.raises(KeyError):                   # reference: torch.load(...)[StyleGAN_pth_key]
        ood_faceGAN_e4e(out_size=1024, build_encoder=False, StyleGAN_pth=kw['StyleGAN_pth'], StyleGAN_pth_key='params_ema')
    torch.save({'state_dict': {'encoder.input_layer.0.weight': torch.zeros(64, 3, 3, 3)}}, tmp_path / 'bad.pt')   # 620 keys missing
    with pytest.raises(RuntimeError):               # encoder.load_state_dict(..., strict=True)
        ood_faceGAN_e4e(out_size=1024, E4E_pth=str(tmp_path / 'bad.pt'))


@pytest.mark.gpu
def test_model_built_from_checkpoints_runs_like_the_strict_loaded_one(tmp_path_factory):
    """forward from an image (encoder + generator + SAMM) of the constructor-loaded model == the strict-loaded model."""
    from oodgan.arch import ood_faceGAN_e4e
    from oodgan.io import load_network_g
    dev = torch.device('cuda:0')
    ref, sd, kw, tmp_path = _recipe(tmp_path_factory, 'basicsr')
    m = ood_faceGAN_e4e(out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True, warp_scale=0.08, cycle_align=2,
                        blend_with_gen=True, ModSize=256, **kw)
    # the filtered SAMM checkpoint of the CLI (modulation.* / feats_conv.* only)
    torch.save({'params_ema': OrderedDict((k, v) for k, v in sd.items() if k.startswith(('modulation.', 'feats_conv.')))}, tmp_path / 'net_g.pth')
    res = load_network_g(m, str(tmp_path / 'net_g.pth'))
    assert not res.unexpected_keys
    m.delta_latent.data = sd['delta_latent'].clone()          # load_network_g resets it, as the CLI does
    for net in (m, ref):
        net.to(dev).eval()
    x = synth.make_images(1024, 1, seed=34).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(1024, 1, seed=35)]
    out_a, lat_a = m(x, noise=noises)
    out_b, lat_b = ref(x, noise=noises)
    assert torch.equal(lat_a, lat_b) and torch.equal(out_a, out_b)
