"""Parameter containers of the encoders that run BEFORE the accelerated path.

Same constructor arguments, state-dict keys (621 entries for the IR-SE-50 e4e encoder), ``channels`` /
``progressive_stage`` attributes as the reference classes —
  ``Encoder4Editing``            src/ops/e4e/encoders/psp_encoders.py:125-216 (IR-SE-50 helpers.py:33-57,60-76,479-501)
  ``ProgressiveBackboneEncoder`` src/ops/restyle/restyle_e4e_encoder.py:37-112
  ``fs_encoder_v2``              src/ops/FeatureStyle/feature_style_encoder.py:12-74 (IResNet-50 arcface/iresnet.py:30-61)
— and NOTHING ELSE: these classes hold parameters (``torch.nn`` modules are used as containers only) and have no forward.
The forward passes live in ``oodgan.encoder_hip`` (``Encoder4EditingHIP`` ... subclass these containers and run every conv,
normalisation, gate and resize through ``liboodgan_hip.so``).  A plain-torch restatement of the same graphs exists only as
test infrastructure (``tests/torch_encoder_mirror.py``): the product package has no PyTorch / MIOpen compute path."""
import math
from collections import namedtuple
from enum import Enum

import numpy as np
import torch
from torch import nn


class ProgressiveStage(Enum):
    WTraining = 0
    Delta1Training = 1
    Delta2Training = 2
    Delta3Training = 3
    Delta4Training = 4
    Delta5Training = 5
    Delta6Training = 6
    Delta7Training = 7
    Delta8Training = 8
    Delta9Training = 9
    Delta10Training = 10
    Delta11Training = 11
    Delta12Training = 12
    Delta13Training = 13
    Delta14Training = 14
    Delta15Training = 15
    Delta16Training = 16
    Delta17Training = 17
    Inference = 18


class _Container(nn.Module):
    """parameter container: the forward pass is implemented by the HIP subclass in oodgan.encoder_hip"""

    def forward(self, *a, **k):
        raise RuntimeError(f'{type(self).__name__} is a parameter container; instantiate oodgan.encoder_hip.{type(self).__name__}HIP '
                           '(the encoders run on the HIP kernels only: there is no PyTorch fallback)')


_Unit = namedtuple('_Unit', ['in_channel', 'depth', 'stride'])


def _stage(in_channel, depth, num_units, stride=2):
    return [_Unit(in_channel, depth, stride)] + [_Unit(depth, depth, 1) for _ in range(num_units - 1)]


def get_blocks(num_layers):
    """helpers.py:33-57."""
    table = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}
    if num_layers not in table:
        raise ValueError(f'Invalid number of layers: {num_layers}. Must be one of [50, 100, 152]')
    n = table[num_layers]
    return [_stage(64, 64, n[0]), _stage(64, 128, n[1]), _stage(128, 256, n[2]), _stage(256, 512, n[3])]


class SEModule(nn.Module):
    """Squeeze-excitation gate (helpers.py:60-76)."""

    def __init__(self, channels, reduction):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.fc2 = nn.Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)


class bottleneck_IR_SE(nn.Module):
    """helpers.py:479-501: BN -> conv3x3 -> PReLU -> conv3x3(stride) -> BN -> SE, plus identity / 1x1+BN shortcut."""

    def __init__(self, in_channel, depth, stride, bn=True):
        super().__init__()
        if in_channel == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(in_channel, depth, (1, 1), stride, bias=False), nn.BatchNorm2d(depth))
        self.res_layer = nn.Sequential(nn.BatchNorm2d(in_channel), nn.Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False),
                                       nn.PReLU(depth), nn.Conv2d(depth, depth, (3, 3), stride, 1, bias=False),
                                       nn.BatchNorm2d(depth), SEModule(depth, 16))


class bottleneck_IR(nn.Module):
    def __init__(self, in_channel, depth, stride, bn=True):
        super().__init__()
        if in_channel == depth:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(in_channel, depth, (1, 1), stride, bias=False), nn.BatchNorm2d(depth))
        self.res_layer = nn.Sequential(nn.BatchNorm2d(in_channel), nn.Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False),
                                       nn.PReLU(depth), nn.Conv2d(depth, depth, (3, 3), stride, 1, bias=False),
                                       nn.BatchNorm2d(depth))


class _EqualLinear(nn.Module):
    """EqualLinear of src/ops/StyleGAN/modules.py:136-170 (no activation on this path)."""

    def __init__(self, in_dim, out_dim, lr_mul=1):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim))
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul


class GradualStyleBlock(nn.Module):
    """psp_encoders.py:35-57: log2(spatial) stride-2 convs + LeakyReLU(0.01), then EqualLinear."""

    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c, self.spatial = out_c, spatial
        mods = [nn.Conv2d(in_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        for _ in range(int(np.log2(spatial)) - 1):
            mods += [nn.Conv2d(out_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        self.convs = nn.Sequential(*mods)
        self.linear = _EqualLinear(out_c, out_c, lr_mul=1)


class Encoder4Editing(_Container):
    def __init__(self, num_layers, mode='ir', opts=None, bn=True):
        super().__init__()
        assert num_layers in [50, 100, 152], 'num_layers should be 50,100, or 152'
        assert mode in ['ir', 'ir_se'], 'mode should be ir or ir_se'
        unit = bottleneck_IR if mode == 'ir' else bottleneck_IR_SE
        self.input_layer = nn.Sequential(nn.Conv2d(3, 64, (3, 3), 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
        self.channels = [64]
        mods = []
        for block in get_blocks(num_layers):
            for u in block:
                mods.append(unit(u.in_channel, u.depth, u.stride, bn=bn))
            self.channels.append(block[-1].depth)
        self.body = nn.Sequential(*mods)
        size = opts.stylegan_size if hasattr(opts, 'stylegan_size') else opts['stylegan_size']
        self.style_count = 2 * int(math.log(size, 2)) - 2
        self.coarse_ind, self.middle_ind = 3, 7
        self.styles = nn.ModuleList()
        for i in range(self.style_count):
            spatial = 16 if i < self.coarse_ind else (32 if i < self.middle_ind else 64)
            self.styles.append(GradualStyleBlock(512, 512, spatial))
        self.latlayer1 = nn.Conv2d(256, 512, kernel_size=1, stride=1, padding=0)
        self.latlayer2 = nn.Conv2d(128, 512, kernel_size=1, stride=1, padding=0)
        self.progressive_stage = ProgressiveStage.Inference

    def get_deltas_starting_dimensions(self):
        return list(range(self.style_count))

    def set_progressive_stage(self, new_stage):
        self.progressive_stage = new_stage


class ProgressiveBackboneEncoder(_Container):
    """ReStyle's encoder (reference src/ops/restyle/restyle_e4e_encoder.py:37-112): the same IR-SE-50 trunk on
    ``opts.input_nc`` input channels (6: image + current reconstruction); all ``n_styles`` codes come from
    ``GradualStyleBlock(512, 512, 16)`` heads on the final 16x16 map (no FPN).  Same state-dict keys, ``channels`` and
    ``progressive_stage`` as the reference (forward: ``encoder_hip.ProgressiveBackboneEncoderHIP``)."""

    def __init__(self, num_layers, mode='ir', n_styles=18, opts=None):
        super().__init__()
        assert num_layers in [50, 100, 152], 'num_layers should be 50,100, or 152'
        assert mode in ['ir', 'ir_se'], 'mode should be ir or ir_se'
        unit = bottleneck_IR if mode == 'ir' else bottleneck_IR_SE
        input_nc = opts.input_nc if hasattr(opts, 'input_nc') else opts['input_nc']
        self.input_layer = nn.Sequential(nn.Conv2d(input_nc, 64, (3, 3), 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
        self.channels = [64]
        mods = []
        for block in get_blocks(num_layers):
            for u in block:
                mods.append(unit(u.in_channel, u.depth, u.stride))
            self.channels.append(block[-1].depth)
        self.body = nn.Sequential(*mods)
        self.style_count = n_styles
        self.styles = nn.ModuleList([GradualStyleBlock(512, 512, 16) for _ in range(n_styles)])
        self.progressive_stage = ProgressiveStage.Inference

    def get_deltas_starting_dimensions(self):
        return list(range(self.style_count))

    def set_progressive_stage(self, new_stage):
        self.progressive_stage = new_stage


class IBasicBlock(nn.Module):
    """Pre-activation basic block of the ArcFace IResNet (reference src/ops/FeatureStyle/arcface/iresnet.py:30-61):
    bn1 -> conv3x3 -> bn2 -> PReLU -> conv3x3(stride) -> bn3, + identity (1x1 conv stride + bn when the shape changes)."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout, eps=1e-5)
        self.prelu = nn.PReLU(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, stride, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout, eps=1e-5)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout, eps=1e-5))
        self.stride = stride


def _iresnet_stage(cin, cout, n):
    return nn.Sequential(IBasicBlock(cin, cout, 2), *[IBasicBlock(cout, cout, 1) for _ in range(n - 1)])


class fs_encoder_v2(_Container):
    """Feature-Style encoder (reference src/ops/FeatureStyle/feature_style_encoder.py:12-74): IResNet-50 trunk (stem +
    stages of 3/4/14/3 blocks), 3x3 adaptive-average-pooled descriptors of the four stages (64+128+256+512 channels x 9)
    -> ``n_styles`` Linear(8640, 512) heads; ``content_layer`` on the 256-channel stage; the stem and the first three
    stages are the SAMM taps.  Same state-dict keys.  The reference builds the trunk from an ArcFace checkpoint
    (``opts.arcface_model_path``) and then overwrites every parameter from ``FeatureStyle_pth``; the mirror only needs the
    latter."""

    def __init__(self, n_styles=18, opts=None, stride=(1, 1), **kwargs):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64, eps=1e-5), nn.PReLU(64))
        self.block_1 = _iresnet_stage(64, 64, 3)
        self.block_2 = _iresnet_stage(64, 128, 4)
        self.block_3 = _iresnet_stage(128, 256, 14)
        self.block_4 = _iresnet_stage(256, 512, 3)
        self.content_layer = nn.Sequential(nn.BatchNorm2d(256), nn.Conv2d(256, 512, 3, 1, 1, bias=False), nn.BatchNorm2d(512),
                                           nn.PReLU(512), nn.Conv2d(512, 512, 3, stride, 1, bias=False), nn.BatchNorm2d(512))
        self.avg_pool = nn.AdaptiveAvgPool2d((3, 3))
        self.styles = nn.ModuleList([nn.Linear(960 * 9, 512) for _ in range(n_styles)])
