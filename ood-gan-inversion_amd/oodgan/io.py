"""Checkpoint plumbing: rosinality ``g_ema`` and BasicSR ``params_ema`` generator
layouts (reference OOD_faceGAN_e4e_arch.py:137-139, BasicSR/scripts/model_conversion/convert_stylegan.py)."""
import torch

from .modules import basicsr_to_rosinality_key


def load_generator_checkpoint(generator, path, key='params_ema'):
    ckpt = torch.load(path, map_location='cpu')[key]       # KeyError for a wrong key, as the reference (:138)
    if any(k.startswith('style_mlp.') or k.startswith('style_conv1.') for k in ckpt):
        ckpt = {basicsr_to_rosinality_key(k): v for k, v in ckpt.items()}
    return generator.load_state_dict(ckpt, strict=False)


def load_network_g(model, path, key='params_ema', strict=False):
    """The CLI's load_model (run_ood_faceGAN_inversion.py:30-47): take ``ckpt[key]``, keep a ``delta_latent`` entry
    only if it is the full (>=3-D) tensor, load (non-strict by default: the SAMM checkpoint holds ``modulation.*`` and
    ``feats_conv.*`` only), then reset ``delta_latent`` to zero."""
    from collections import OrderedDict
    ckpt = torch.load(path, map_location='cpu')
    if key is not None:
        ckpt = ckpt[key]
    kept = OrderedDict((k, v) for k, v in ckpt.items() if 'delta_latent' not in k or v.dim() >= 3)
    res = model.load_state_dict(kept, strict=strict)
    model.delta_latent.data = torch.zeros_like(model.delta_latent)
    return res


def load_direction(directions_dir, editing):
    """Editing direction of a dataset block (run_ood_faceGAN_inversion.py:49-62): ``<dir>/<direction>.npy`` x intensity,
    shape (1, ...); ``None`` -> scalar zero."""
    import os
    import numpy as np
    if editing is None:
        return torch.tensor(0.)
    d = np.load(os.path.join(directions_dir, editing['direction'] + '.npy'))
    return torch.tensor(d, dtype=torch.float32).unsqueeze(0) * editing['intensity']
