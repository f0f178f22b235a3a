"""Checkpoint plumbing: rosinality ``g_ema`` and BasicSR ``params_ema`` generator
layouts (reference OOD_faceGAN_e4e_arch.py:137-139, BasicSR/scripts/model_conversion/convert_stylegan.py)."""
import torch

from .modules import basicsr_to_rosinality_key


def load_generator_checkpoint(generator, path, key='params_ema'):
    ckpt = torch.load(path, map_location='cpu')
    if isinstance(ckpt, dict) and key in ckpt:
        ckpt = ckpt[key]
    if any(k.startswith('style_mlp.') or k.startswith('style_conv1.') for k in ckpt):
        ckpt = {basicsr_to_rosinality_key(k): v for k, v in ckpt.items()}
    return generator.load_state_dict(ckpt, strict=False)
