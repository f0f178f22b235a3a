"""Deterministic synthetic parameters / inputs for the hot path.

There is no network on the build or GPU boxes, so the pretrained checkpoints the
reference loads (rosinality ``stylegan2-ffhq-config-f.pth['g_ema']``, the filtered SAMM
checkpoint, reference OOD_faceGAN_e4e_arch.py:137-153) are absent.  Every tensor is instead
drawn from a numpy PCG64 stream keyed by the *state-dict key name* and a seed, so the same
values can be rebuilt (a) inside the golden-vector generator that fills the real reference
modules, (b) in the CPU oracle tests, (c) on the GPU box for parity tests and ``bench.py``.

Key names follow the reference state dicts: rosinality ``Generator``
layout (reference src/ops/StyleGAN/model.py:375-459) and the ``ood_faceGAN_e4e`` layout
(``generator.*``, ``modulation.{i}.alignment.body.body.*``, ``feats_conv.{i}.*``).
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

__all__ = [
    'encoder_state',
    'normal', 'uniform', 'make_kernel', 'generator_channels', 'generator_state', 'samm_state',
    'ood_state', 'make_noises', 'make_latents', 'make_images', 'make_encoder_feats', 'lpips_state',
]


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode('utf-8')), int(seed) & 0x7FFFFFFF])


def normal(name, shape, seed=0, std=1.0, mean=0.0):
    """float32 tensor ~ N(mean, std^2), reproducible from (name, seed) alone."""
    a = _rng(name, seed).standard_normal(size=tuple(shape), dtype=np.float64)
    return torch.from_numpy((a * std + mean).astype(np.float32))


def uniform(name, shape, seed=0, lo=0.0, hi=1.0):
    a = _rng(name, seed).random(size=tuple(shape), dtype=np.float64)
    return torch.from_numpy((a * (hi - lo) + lo).astype(np.float32))


def make_kernel(taps=(1, 3, 3, 1)):
    """outer(k,k)/sum — reference src/ops/StyleGAN/model.py:19-27."""
    k = torch.tensor(taps, dtype=torch.float32)
    k2 = k[None, :] * k[:, None]
    return k2 / k2.sum()


def generator_channels(channel_multiplier=2, narrow=1):
    """Resolution -> channels (reference model.py:402-412)."""
    return {
        4: int(512 * narrow), 8: int(512 * narrow), 16: int(512 * narrow), 32: int(512 * narrow),
        64: int(256 * channel_multiplier * narrow), 128: int(128 * channel_multiplier * narrow),
        256: int(64 * channel_multiplier * narrow), 512: int(32 * channel_multiplier * narrow),
        1024: int(16 * channel_multiplier * narrow),
    }


def _styled_conv(sd, prefix, cin, cout, style_dim, seed, upsample, noise_w):
    sd[f'{prefix}.conv.weight'] = normal(f'{prefix}.conv.weight', (1, cout, cin, 3, 3), seed)
    if upsample:
        sd[f'{prefix}.conv.blur.kernel'] = make_kernel() * 4.0
    sd[f'{prefix}.conv.modulation.weight'] = normal(f'{prefix}.conv.modulation.weight', (cin, style_dim), seed)
    sd[f'{prefix}.conv.modulation.bias'] = normal(f'{prefix}.conv.modulation.bias', (cin,), seed, 0.1, 1.0)
    sd[f'{prefix}.noise.weight'] = torch.full((1,), float(noise_w)) + normal(f'{prefix}.noise.weight', (1,), seed, 0.01)
    sd[f'{prefix}.activate.bias'] = normal(f'{prefix}.activate.bias', (cout,), seed, 0.1)


def _to_rgb(sd, prefix, cin, style_dim, seed, upsample):
    sd[f'{prefix}.bias'] = normal(f'{prefix}.bias', (1, 3, 1, 1), seed, 0.1)
    if upsample:
        sd[f'{prefix}.upsample.kernel'] = make_kernel() * 4.0
    sd[f'{prefix}.conv.weight'] = normal(f'{prefix}.conv.weight', (1, 3, cin, 1, 1), seed)
    sd[f'{prefix}.conv.modulation.weight'] = normal(f'{prefix}.conv.modulation.weight', (cin, style_dim), seed)
    sd[f'{prefix}.conv.modulation.bias'] = normal(f'{prefix}.conv.modulation.bias', (cin,), seed, 0.1, 1.0)


def generator_state(size, style_dim=512, n_mlp=8, channel_multiplier=2, seed=0, lr_mlp=0.01,
                    noise_weight=0.1, prefix='', narrow=1, blur_kernel=(1, 3, 3, 1), upsample_kernel=(1, 3, 3, 1)):
    """Full rosinality-layout state dict for ``Generator(size, style_dim, n_mlp, cm, blur_kernel)``.

    Init distributions follow the reference constructors (randn conv / modulation weights,
    ``randn/lr_mul`` mapping weights, modulation bias 1: model.py:129-158,219-223) except that
    biases and noise strengths are made non-zero so that every term is exercised and the
    OOD callback's division by ``NoiseInjection.weight`` is finite.
    """
    ch = generator_channels(channel_multiplier, narrow)
    log_size = int(math.log2(size))
    sd = OrderedDict()
    for i in range(1, n_mlp + 1):
        sd[f'style.{i}.weight'] = normal(f'style.{i}.weight', (style_dim, style_dim), seed, 1.0 / lr_mlp)
        sd[f'style.{i}.bias'] = normal(f'style.{i}.bias', (style_dim,), seed, 1.0)
    sd['input.input'] = normal('input.input', (1, ch[4], 4, 4), seed)
    _styled_conv(sd, 'conv1', ch[4], ch[4], style_dim, seed, False, noise_weight)
    _to_rgb(sd, 'to_rgb1', ch[4], style_dim, seed, False)
    cin = ch[4]
    k = 0
    for i in range(3, log_size + 1):
        cout = ch[2 ** i]
        _styled_conv(sd, f'convs.{k}', cin, cout, style_dim, seed, True, noise_weight)
        _styled_conv(sd, f'convs.{k + 1}', cout, cout, style_dim, seed, False, noise_weight)
        _to_rgb(sd, f'to_rgbs.{k // 2}', cout, style_dim, seed, True)
        cin = cout
        k += 2
    num_layers = (log_size - 2) * 2 + 1
    for li in range(num_layers):
        r = 2 ** ((li + 5) // 2)
        sd[f'noises.noise_{li}'] = normal(f'noises.noise_{li}', (1, 1, r, r), seed)
    # the registered buffers of Blur (from the constructor's ``blur_kernel``) and of ToRGB's Upsample (model.py:455: always [1,3,3,1] as
    # constructed; a checkpoint may hold anything) — model.py:30-48,72-81
    for k_ in sd:
        if k_.endswith('.blur.kernel') and tuple(blur_kernel) != (1, 3, 3, 1):
            sd[k_] = make_kernel(blur_kernel) * 4.0
        if k_.endswith('.upsample.kernel') and tuple(upsample_kernel) != (1, 3, 3, 1):
            sd[k_] = make_kernel(upsample_kernel) * 4.0
    if prefix:
        sd = OrderedDict((prefix + k_, v) for k_, v in sd.items())
    return sd


def _bottleneck(sd, prefix, cin, depth, seed):
    """Keys of ``bottleneck_IR(cin, depth, 1, bn='InstanceNorm', bias=False)``
    (reference src/ops/e4e/encoders/helpers.py:426-448)."""
    if cin != depth:
        n = f'{prefix}.shortcut_layer.0.weight'
        sd[n] = normal(n, (depth, cin, 1, 1), seed, math.sqrt(2.0 / (cin + depth)))
        n = f'{prefix}.shortcut_layer.1.weight'
        sd[n] = normal(n, (depth,), seed, 0.1, 1.0)
        n = f'{prefix}.shortcut_layer.1.bias'
        sd[n] = normal(n, (depth,), seed, 0.1)
    n = f'{prefix}.res_layer.0.weight'
    sd[n] = normal(n, (cin,), seed, 0.1, 1.0)
    n = f'{prefix}.res_layer.0.bias'
    sd[n] = normal(n, (cin,), seed, 0.1)
    n = f'{prefix}.res_layer.1.weight'
    sd[n] = normal(n, (depth, cin, 3, 3), seed, math.sqrt(2.0 / (9 * (cin + depth))))
    n = f'{prefix}.res_layer.2.weight'
    sd[n] = normal(n, (depth,), seed, 0.05, 0.25)
    n = f'{prefix}.res_layer.3.weight'
    sd[n] = normal(n, (depth, depth, 3, 3), seed, math.sqrt(2.0 / (9 * (depth + depth))))
    n = f'{prefix}.res_layer.4.weight'
    sd[n] = normal(n, (depth,), seed, 0.1, 1.0)
    n = f'{prefix}.res_layer.4.bias'
    sd[n] = normal(n, (depth,), seed, 0.1)


def samm_state(chn, prefix, seed=0):
    """State of one ``StyledscaleNshfitBlock(chn, chn, btn=None)`` (reference
    src/ops/SAMM/helpers.py:182-216): an ``SPM_Warp`` whose ``AlignNet`` body is two
    ``bottleneck_IR`` on 2*chn channels (xavier-normal conv weights, helpers.py:124-127)."""
    sd = OrderedDict()
    sd[f'{prefix}.weight'] = torch.ones(1)
    _bottleneck(sd, f'{prefix}.alignment.body.body.0', 2 * chn, 2 * chn, seed)
    _bottleneck(sd, f'{prefix}.alignment.body.body.1', 2 * chn, 3, seed)
    sd[f'{prefix}.alignment.blur.kernel'] = make_kernel()
    sd[f'{prefix}.noiseInj.weight'] = torch.zeros(1)
    return sd


def ood_state(out_size=1024, style_dim=512, n_mlp=8, channel_multiplier=2, seed=0,
              enc_channels=(64, 64, 128, 256), noise_weight=0.1):
    """State of ``ood_faceGAN_e4e`` minus the e4e encoder (``encoder.*`` keys): generator,
    4 SAMM blocks for 256,128,64,32 px (reference OOD_faceGAN_e4e_arch.py:108-116), the four
    1x1 ``feats_conv`` (:70-75), ``avg_latent`` and ``delta_latent`` (:124-129)."""
    ch = generator_channels(channel_multiplier)
    sd = OrderedDict()
    sd['avg_latent'] = normal('avg_latent', (1, style_dim), seed, 0.5)
    sd['delta_latent'] = normal('delta_latent', (1, int(math.log2(out_size)) * 2 - 2, style_dim), seed, 0.05)
    featsize = 256
    for i in range(4):
        cout = ch[featsize]
        n = f'feats_conv.{i}.weight'
        sd[n] = normal(n, (cout, enc_channels[i], 1, 1), seed, 1.0 / math.sqrt(enc_channels[i]))
        n = f'feats_conv.{i}.bias'
        sd[n] = normal(n, (cout,), seed, 0.1)
        featsize //= 2
    for j, i in enumerate(range(8, 4, -1)):
        sd.update(samm_state(ch[2 ** i], f'modulation.{j}', seed))
    sd.update(generator_state(out_size, style_dim, n_mlp, channel_multiplier, seed,
                              noise_weight=noise_weight, prefix='generator.'))
    return sd


def make_noises(size, batch, seed=2):
    """17 (for 1024) per-layer noise maps (B,1,r,r), r = 4,8,8,16,16,...."""
    log_size = int(math.log2(size))
    out = []
    for li in range((log_size - 2) * 2 + 1):
        r = 2 ** ((li + 5) // 2)
        out.append(normal(f'noise_in.{li}', (batch, 1, r, r), seed))
    return out


def make_latents(size, batch, style_dim=512, seed=3, std=1.0):
    return normal('latent_in', (batch, int(math.log2(size)) * 2 - 2, style_dim), seed, std)


def make_images(size, batch, seed=1):
    return normal('image_in', (batch, 3, size, size), seed).clamp_(-1.0, 1.0)


def make_encoder_feats(batch, seed=4, channels=(64, 64, 128, 256), sizes=(256, 128, 64, 32)):
    """Stand-ins for the e4e feature pyramid taps (reference psp_encoders.py:186-214)."""
    return [normal(f'enc_feat.{i}', (batch, c, s, s), seed) for i, (c, s) in enumerate(zip(channels, sizes))]


def encoder_state(shapes, seed=0, prefix=''):
    """Deterministic parameters for the e4e encoder from its (key -> shape) table (``{k: v.shape for k, v in
    Encoder4Editing(...).state_dict().items()}``): He-normal conv weights, BatchNorm affine/statistics near
    identity (running_var in [0.5,1.5]), PReLU slopes ~0.25, EqualLinear weights N(0,1)."""
    sd = OrderedDict()
    for k, shp in shapes.items():
        shp = tuple(shp)
        name = prefix + k
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('running_mean'):
            sd[k] = normal(name, shp, seed, 0.1)
        elif k.endswith('running_var'):
            sd[k] = uniform(name, shp, seed, 0.5, 1.5)
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            sd[k] = normal(name, shp, seed, math.sqrt(2.0 / fan_in))
        elif len(shp) == 2:
            sd[k] = normal(name, shp, seed, 1.0)
        elif k.endswith('.bias'):
            sd[k] = normal(name, shp, seed, 0.1)
        elif '.res_layer.2.' in k or k.startswith('input_layer.2.'):
            sd[k] = normal(name, shp, seed, 0.05, 0.25)           # PReLU slopes
        elif '.res_layer.4.weight' in k:
            # last BatchNorm of the residual branch: small gain, so the 24 un-normalised residual units stay
            # O(1) and well-conditioned (with gain 1 the net amplifies a 1e-6 input perturbation to O(1))
            sd[k] = normal(name, shp, seed, 0.03, 0.25)
        else:
            sd[k] = normal(name, shp, seed, 0.1, 1.0)             # BatchNorm weight
    return sd


def restyle_checkpoint(seed=0, style_cnt=18, style_dim=512):
    """A ReStyle-e4e checkpoint in the layout the reference loads (OOD_faceGAN_restyle_arch.py:67-83): ``state_dict``
    with ``encoder.*`` keys (ProgressiveBackboneEncoder, 6 input channels), ``latent_avg`` (style_cnt, style_dim), ``opts``."""
    from .encoder import ProgressiveBackboneEncoder
    opts = {'encoder_type': 'ProgressiveBackboneEncoder', 'input_nc': 6}
    shapes = {k: tuple(v.shape) for k, v in ProgressiveBackboneEncoder(50, 'ir_se', style_cnt, opts).state_dict().items()}
    enc = encoder_state(shapes, seed=seed, prefix='restyle.')
    for k in enc:
        if k.endswith('.linear.weight'):
            enc[k] = enc[k] / 256.0          # recipe trunk features are O(10): keep the predicted codes O(1) like trained ones
    return {'state_dict': OrderedDict(('encoder.' + k, v) for k, v in enc.items()),
            'latent_avg': normal('restyle.latent_avg', (style_cnt, style_dim), seed, 0.5), 'opts': opts}


def featurestyle_state(seed=0, style_cnt=18):
    """State dict of ``fs_encoder_v2`` (the content of ``FeatureStyle_pth``, OOD_faceGAN_featureStyle_arch.py:74-79)."""
    from .encoder import fs_encoder_v2
    shapes = {k: tuple(v.shape) for k, v in fs_encoder_v2(style_cnt, stride=(2, 2)).state_dict().items()}
    sd = OrderedDict()
    for k, shp in shapes.items():
        name = 'fs.' + k
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('running_mean'):
            sd[k] = normal(name, shp, seed, 0.1)
        elif k.endswith('running_var'):
            sd[k] = uniform(name, shp, seed, 0.5, 1.5)
        elif len(shp) == 4:
            sd[k] = normal(name, shp, seed, math.sqrt(2.0 / (shp[1] * shp[2] * shp[3])))
        elif len(shp) == 2:
            sd[k] = normal(name, shp, seed, 0.1 / math.sqrt(shp[1]))      # Linear heads: recipe descriptors are O(10), codes O(1)
        elif 'prelu' in k or k in ('conv.2.weight', 'content_layer.3.weight'):
            sd[k] = normal(name, shp, seed, 0.05, 0.25)                     # PReLU slopes
        elif k.endswith('.bias'):
            sd[k] = normal(name, shp, seed, 0.1)
        elif '.bn3.weight' in k:
            sd[k] = normal(name, shp, seed, 0.03, 0.25)                     # damp the residual branch (see encoder_state)
        else:
            sd[k] = normal(name, shp, seed, 0.1, 1.0)
    return sd


def lpips_state(seed=0):
    """Seeded stand-in for the state dict of ``lpips.LPIPS(net='alex')`` (key names of lpips 0.1.x: the torchvision AlexNet feature convs
    under ``net.slice{1..5}.{0,3,6,8,10}`` and the 1x1 ``lin{0..4}.model.1.weight`` layers, which the package keeps non-negative).  The
    pretrained weights are absent: He-scaled normals keep the five taps O(1) so that every layer contributes to the loss."""
    sd = OrderedDict()
    for (sl, idx), (co, ci, k) in zip(((1, 0), (2, 3), (3, 6), (4, 8), (5, 10)), ((64, 3, 11), (192, 64, 5), (384, 192, 3), (256, 384, 3), (256, 256, 3))):
        sd[f'net.slice{sl}.{idx}.weight'] = normal(f'lpips.w{sl}', (co, ci, k, k), seed, math.sqrt(2.0 / (ci * k * k)))
        sd[f'net.slice{sl}.{idx}.bias'] = normal(f'lpips.b{sl}', (co,), seed, 0.1)
    for k, c in enumerate((64, 192, 384, 256, 256)):
        sd[f'lin{k}.model.1.weight'] = normal(f'lpips.lin{k}', (1, c, 1, 1), seed, 1.0).abs() * (4.0 / c)
    return sd
