"""Host-side mirror of the reference's StyleGAN2 building blocks (same class names, constructor
arguments, forward signatures and state-dict keys), with every forward routed to the HIP kernels.

reference: src/ops/StyleGAN/model.py (rosinality flavour — the one the OOD archs instantiate) and
src/ops/StyleGAN/stylegan2_arch.py (BasicSR flavour ``StyleGAN2Generator``).  ``torch.nn.Module`` is
used only as the parameter container (state_dict / load_state_dict / cuda / eval / named_parameters).
Weights are treated as frozen: no gradient w.r.t. parameters is produced (the path optimises
latents only)."""
import math
import random

import os

import torch
from torch import nn

from . import ops
from .engine import GeneratorEngine

# model(x): carry the forward range scales from call to call (the W+ loop's scheme: no measurement pass per conv input, fused producers; B = 8 forward
# 4 % faster, B = 1 1-2 %).  OFF by default since round 6: with carried scales a call can differ from a cold call at fp32-rounding level (<= 6e-6 on the
# image) depending on what ran before — the reference's forward is a pure function of its inputs, and so is this one by default.  Opt in with
# OODGAN_CARRY_FORWARD=1 (or ``oodgan.modules.CARRY_FORWARD = True``); the W+ loop's own carried scales (engine._WRun) are not affected.
CARRY_FORWARD = os.environ.get('OODGAN_CARRY_FORWARD', '0') == '1'
from .synth import generator_channels, make_kernel

__all__ = ['PixelNorm', 'EqualLinear', 'ModulatedConv2d', 'NoiseInjection', 'ConstantInput', 'StyledConv', 'ToRGB',
           'Upsample', 'Blur', 'FusedLeakyReLU', 'Generator', 'StyleGAN2Generator', 'upfirdn2d', 'fused_leaky_relu',
           'make_kernel']

upfirdn2d = ops.upfirdn2d
fused_leaky_relu = ops.fused_leaky_relu


class FusedLeakyReLU(nn.Module):
    """reference src/ops/op/fused_act.py:79-89."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5, device='hip'):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope, self.scale, self.device = negative_slope, scale, device

    def forward(self, input):
        return ops.fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


class PixelNorm(nn.Module):
    def forward(self, input):
        return ops.pixel_norm(input)


class Upsample(nn.Module):
    """reference model.py:30-48."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        k = make_kernel(kernel) * (factor ** 2)
        self.register_buffer('kernel', k)
        p = k.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return ops.upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Blur(nn.Module):
    """reference model.py:72-88."""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        k = make_kernel(kernel)
        if upsample_factor > 1:
            k = k * (upsample_factor ** 2)
        self.register_buffer('kernel', k)
        self.pad = pad

    def forward(self, input):
        return ops.upfirdn2d(input, self.kernel, pad=self.pad)


class EqualLinear(nn.Module):
    """reference model.py:129-158."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        return ops.equal_linear(input, self.weight, self.bias, self.lr_mul, bool(self.activation))


class ModulatedConv2d(nn.Module):
    """reference model.py:178-274 (plain / upsample branches; ``downsample`` is not on the path)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if downsample:
            raise NotImplementedError('downsample ModulatedConv2d is discriminator-only (off the hot path)')
        if kernel_size not in (1, 3):
            raise NotImplementedError('kernel_size must be 1 or 3')
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample, self.demodulate = upsample, downsample, demodulate
        if upsample:
            p = (len(blur_kernel) - 2) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + 1, p // 2 + 1), upsample_factor=2)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self._prep_key, self._prep = None, None

    def _prepared(self):
        key = (self.weight.data_ptr(), self.weight._version, str(self.weight.device))
        if key != self._prep_key:
            w = self.weight.detach()[0]
            prep = {}
            if self.kernel_size == 3:
                prep['wpk'] = ops.pack_conv3x3(w, self.scale)
                prep['wsq'] = ops.weight_sqsum(w)
            self._prep_key, self._prep = key, prep
        return self._prep

    def forward(self, input, style):
        s = self.modulation(style)                       # (B, Ci)
        if self.kernel_size == 1:
            if self.demodulate or self.out_channel != 3:
                raise NotImplementedError('1x1 modulated conv is implemented for ToRGB (3 outputs, no demod)')
            return ops.torgb(input, self.weight.detach().reshape(3, -1), s)
        prep = self._prepared()
        d = ops.demod(s, prep['wsq'], self.scale) if self.demodulate else None
        if self.upsample:
            z = ops.conv3x3(input, prep['wpk'], self.out_channel, ops.CONV_T2, in_scale=s, out_scale=d)
            H2, W2 = 2 * input.shape[2] + 1, 2 * input.shape[3] + 1
            return ops.blur_bias_act(z, self.blur.kernel, self.blur.pad, act=False, in_hw=(H2, W2), in_pitch=z.shape[3])
        return ops.conv3x3(input, prep['wpk'], self.out_channel, ops.CONV_S1, in_scale=s, out_scale=d)


class NoiseInjection(nn.Module):
    """reference model.py:277-292 (incl. the callback hook at :288-290)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None, **kwargs):
        if noise is None:
            b, _, h, w = image.shape
            noise = torch.randn(b, 1, h, w, device=image.device, dtype=image.dtype)
            cb = kwargs.get('callback', None)
            if cb:
                kwargs.update({'noise_weight': self.weight, 'noise': noise})
                noise = cb(image, **kwargs)
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    """reference model.py:308-350: conv -> noise -> bias + lrelu*sqrt2, fused into the conv epilogue
    (plain) or into the blur (upsample) when no callback intervenes."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True, **kwargs):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        # noiseInjection=False / activation=False (model.py:332-341; used by SAMM's styleBlock, helpers.py:43-57): the reference
        # puts a lambda there, so the state dict has no noise.weight / activate.bias either
        self.noise = NoiseInjection() if kwargs.get('noiseInjection', True) else None
        self.activate = FusedLeakyReLU(out_channel) if kwargs.get('activation', True) else None

    def forward(self, input, style, noise=None, **kwargs):
        out = self.conv(input, style)
        if self.noise is None:
            return out if self.activate is None else self.activate(out)
        if self.activate is None:
            kwargs.update({'style': style})
            return self.noise(out, noise=noise, **kwargs)
        if noise is None:
            b, _, h, w = out.shape
            noise = torch.randn(b, 1, h, w, device=out.device, dtype=out.dtype)
            cb = kwargs.get('callback', None)
            if cb:
                kwargs.update({'style': style, 'noise_weight': self.noise.weight, 'noise': noise})
                noise = cb(out, **kwargs)
        return ops.bias_noise_act(out, self.activate.bias, noise, self.noise.weight)


class ToRGB(nn.Module):
    """reference model.py:353-372."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None):
        s = self.conv.modulation(style)
        kern = self.upsample.kernel if skip is not None else None
        return ops.torgb(input, self.conv.weight.detach().reshape(3, -1), s, self.bias.detach().reshape(3), skip, kern)


class Generator(nn.Module):
    """``Generator(size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1,3,3,1], lr_mlp=0.01)``
    reference model.py:375-585; state-dict keys identical."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01, narrow=1, rgb_blur_kernel=None):
        super().__init__()
        if len(blur_kernel) != 4:
            raise NotImplementedError(f'blur_kernel {list(blur_kernel)}: the fused producers are built for four taps (every shipped config: [1,3,3,1])')
        self.size, self.style_dim, self.channel_multiplier, self.narrow = size, style_dim, channel_multiplier, narrow
        self.style = nn.Sequential(PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation='fused_lrelu')
                                                   for _ in range(n_mlp)])
        # ``narrow`` (not a parameter of the reference's model.Generator: the keyword serves StyleGAN2Generator, stylegan2_arch.py:422,435-443) scales
        # every channel count; the matrix kernels take 16-channel blocks — GeneratorEngine zero-pads other counts (engine._pad_channels_to_16)
        self.channels = generator_channels(channel_multiplier, narrow)
        used = [self.channels[2 ** i] for i in range(2, int(math.log(size, 2)) + 1)]
        if any(c < 1 for c in used):
            raise ValueError(f'channel counts {used} (narrow={narrow}, channel_multiplier={channel_multiplier})')
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs, self.upsamples, self.to_rgbs, self.noises = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f'noise_{layer_idx}', torch.randn(1, 1, 2 ** res, 2 ** res))
        cin = self.channels[4]
        for i in range(3, self.log_size + 1):
            cout = self.channels[2 ** i]
            self.convs.append(StyledConv(cin, cout, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(cout, cout, 3, style_dim, blur_kernel=blur_kernel))
            # model.py:455 builds ToRGB without ``blur_kernel`` (its Upsample keeps [1,3,3,1]); stylegan2_arch.py:494 passes resample_kernel: ``rgb_blur_kernel``
            self.to_rgbs.append(ToRGB(cout, style_dim) if rgb_blur_kernel is None else ToRGB(cout, style_dim, blur_kernel=list(rgb_blur_kernel)))
            cin = cout
        self.n_latent = self.log_size * 2 - 2
        self._engine_obj, self._engine_key = None, None

    # -- prepared-weights engine (packed weights, demodulation tables, concatenated style matrix).  Keyed on the storage
    # address and in-place version counter of every parameter and buffer, so that ANY way of changing the weights — .to(),
    # load_state_dict on this module or on a parent (which recurses through _load_from_state_dict and never reaches an
    # override here), optimiser steps, manual copy_ — rebuilds it; encoder_hip._Packed keys its cache the same way.
    def _weights_key(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def engine(self):
        key = self._weights_key()
        if self._engine_obj is None or key != self._engine_key:
            self._engine_obj = GeneratorEngine(self.state_dict(), self.size, self.style_dim, self.channel_multiplier, narrow=self.narrow)
            self._engine_key = key
        return self._engine_obj

    def make_noise(self):
        dev = self.input.input.device
        noises = [torch.randn(1, 1, 4, 4, device=dev)]
        for i in range(3, self.log_size + 1):
            noises += [torch.randn(1, 1, 2 ** i, 2 ** i, device=dev) for _ in range(2)]
        return noises

    def mean_latent(self, n_latent):
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(z).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def _draw_noises(self, batch, noise, randomize_noise):
        if noise is None:
            if randomize_noise:
                noise = [None] * self.num_layers
            else:
                noise = [getattr(self.noises, f'noise_{i}') for i in range(self.num_layers)]
        dev = self.input.input.device
        out = []
        for i, nz in enumerate(noise):
            if nz is None:
                r = 2 ** ((i + 5) // 2)
                nz = torch.randn(batch, 1, r, r, device=dev)
            out.append(nz)
        return out

    def forward(self, styles, return_latents=False, return_features=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, input_is_tensor=False, noise=None, randomize_noise=True,
                conditions=None, cond_layers=None, cond_type='SFT', **kwargs):
        if not input_is_latent and not input_is_tensor:
            styles = [self.style(s) for s in styles]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if not input_is_tensor:
            if len(styles) < 2:
                inject_index = self.n_latent
                latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1) if styles[0].ndim < 3 else styles[0]
            else:
                if inject_index is None:
                    inject_index = random.randint(1, self.n_latent - 1)
                latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                    styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        else:
            latent = styles
        latent = latent.contiguous().float()
        B = latent.shape[0]
        hook = None
        post = None
        cl = None
        if cond_layers is not None and conditions is not None and cond_type != 'NOISE':
            # model.py:561-564: the styled conv runs as usual, then feature_modulation(out, condition, None, cond_type)
            cl = list(cond_layers)
            cb = kwargs.get('callback', None)

            def post(k, out):
                cond = conditions[k]
                if cond_type == 'ADD' and cb is not None:        # feature_modulation's 'ADD' branch asks the callback
                    kw = dict(kwargs)
                    kw.update({'style': latent[:, cl[k]], 'index': k})
                    return ops.feature_modulation(out, [None, cb(out, **kw)], None, 'ADD')
                return ops.feature_modulation(out, cond, None, cond_type)
        elif cond_layers is not None and conditions is not None:
            cl = list(cond_layers)
            direct = kwargs.get('cond_hook', None)
            cb = kwargs.get('callback', None)
            if direct is not None:
                hook = direct
            elif cb is not None:
                def hook(k, raw, style, nz, nw):
                    kw = dict(kwargs)
                    kw.update({'index': k, 'style': style, 'noise_weight': nw, 'noise': nz})
                    return raw + nw * cb(raw, **kw) - nw * nz
            # conditions[k][1] given explicitly as noise (reference model.py:569) replaces the drawn noise
            noise = list(noise) if noise is not None else [None] * self.num_layers
            for k, i in enumerate(cl):
                if conditions[k] is not None and len(conditions[k]) > 1 and conditions[k][1] is not None:
                    noise[i] = conditions[k][1]
        noises = self._draw_noises(B, noise, randomize_noise)
        # the last activation is only materialised in NCHW when the caller asks for it (the engine may keep it in F-form)
        want_feat = bool(return_features) and not return_latents
        eng = self.engine()
        kw = dict(cond_hook=hook, cond_layers=cl, return_features=want_feat, post_hook=post,
                  features_in=kwargs.get('features_in', None), feature_scale=kwargs.get('feature_scale', 1.0))
        # CARRY_FORWARD (opt-in): forward range scales (split-f16 path) carried from the previous forward of this batch size — the W+ loop's scheme
        # (engine.py, ops.FwdRange) for model(x) too: no measurement pass per conv input, the fused producers of the loop.  Scales are powers of two
        # and exact as long as the carried scale keeps the new maximum inside [1, 2^15) — if it does not, the flag is set and the pass
        # is repeated with measured scales.  The first forward of a batch size (measured scales, unfused producers) and the following
        # ones therefore differ by fp32 rounding (different kernels, <= 2e-5 on the image); the following ones are bit-identical among
        # themselves.  Inside a stream capture the flag cannot be read: oodgan.arch.GraphedForward checks it after every replay.
        mode = 'carry' if (CARRY_FORWARD and eng.sform and eng.carry_range) else 'exact'
        res = eng.forward(latent, noises, range_mode=mode, **kw)
        if mode == 'carry' and not torch.cuda.is_current_stream_capturing() and eng.fwd_range_violated():
            eng.reset_fwd_state()
            res = eng.forward(latent, noises, range_mode='exact', **kw)
        image, feat = res if want_feat else (res, None)
        if return_latents:
            return image, latent
        if return_features:
            return image, feat
        return image, None


# BasicSR key layout -> rosinality key layout (BasicSR/scripts/model_conversion/convert_stylegan.py)
def basicsr_to_rosinality_key(k):
    k = k.replace('style_mlp.', 'style.').replace('constant_input.weight', 'input.input')
    k = k.replace('style_conv1.', 'conv1.').replace('style_convs.', 'convs.')
    k = k.replace('.modulated_conv.', '.conv.')
    if k.startswith('noises.noise') and not k.startswith('noises.noise_'):
        k = 'noises.noise_' + k[len('noises.noise'):]
    parts = k.split('.')
    if parts[0] in ('conv1', 'convs') and parts[-1] == 'weight' and 'conv' not in parts[1:-1]:
        k = '.'.join(parts[:-1] + ['noise', 'weight'])     # StyleConv.weight (shape [1]) is the noise strength
    return k


class StyleGAN2Generator(nn.Module):
    """BasicSR-flavour signature (reference src/ops/StyleGAN/stylegan2_arch.py:399-605) over the same
    kernels: parameters are stored under the BasicSR names and remapped to the engine's layout."""

    def __init__(self, out_size, num_style_feat=512, num_mlp=8, channel_multiplier=2, resample_kernel=(1, 3, 3, 1),
                 lr_mlp=0.01, narrow=1):
        super().__init__()
        # narrow (stylegan2_arch.py:422,435-443): every channel count x narrow — built where all of them stay multiples of 16 (round 6)
        self._inner = [Generator(out_size, num_style_feat, num_mlp, channel_multiplier, list(resample_kernel), lr_mlp, narrow=narrow,
                                 rgb_blur_kernel=list(resample_kernel))]
        inner = self._inner[0]
        self.num_style_feat, self.out_size = num_style_feat, out_size
        self.log_size, self.num_layers, self.num_latent = inner.log_size, inner.num_layers, inner.n_latent
        self._names = {}
        for rk, v in inner.state_dict().items():
            if rk.endswith('.kernel'):
                continue                                    # derived FIR constants are not registered in BasicSR
            bk = self._ros_to_basicsr(rk)
            flat = bk.replace('.', '__')
            if rk.startswith('noises.'):
                self.register_buffer(flat, v.clone())
            else:
                self.register_parameter(flat, nn.Parameter(v.clone()))
            self._names[flat] = (bk, rk)

    @staticmethod
    def _ros_to_basicsr(k):
        k = k.replace('style.', 'style_mlp.', 1) if k.startswith('style.') else k
        k = k.replace('input.input', 'constant_input.weight')
        if k.startswith('conv1.'):
            k = 'style_conv1.' + k[len('conv1.'):]
        if k.startswith('convs.'):
            k = 'style_convs.' + k[len('convs.'):]
        k = k.replace('.conv.', '.modulated_conv.').replace('.noise.weight', '.weight')
        if k.startswith('noises.noise_'):
            k = 'noises.noise' + k[len('noises.noise_'):]
        return k

    def state_dict(self, *a, **k):
        from collections import OrderedDict
        return OrderedDict((bk, getattr(self, flat).detach()) for flat, (bk, rk) in self._names.items())

    def load_state_dict(self, sd, strict=True):
        known = {bk: flat for flat, (bk, rk) in self._names.items()}
        missing = [bk for bk in known if bk not in sd]
        unexpected = [k for k in sd if k not in known]
        if strict and (missing or unexpected):
            raise RuntimeError(f'StyleGAN2Generator.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}')
        with torch.no_grad():
            for bk, flat in known.items():
                if bk in sd:
                    getattr(self, flat).copy_(sd[bk])
        self._sync()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._inner[0]._apply(fn)
        self._sync()
        return r

    def _sync(self):
        inner = self._inner[0]
        sd = {rk: getattr(self, flat).detach() for flat, (bk, rk) in self._names.items()}
        inner.load_state_dict(sd, strict=False)

    def make_noise(self):
        return self._inner[0].make_noise()

    def get_latent(self, x):
        return self._inner[0].get_latent(x)

    def mean_latent(self, num_latent):
        return self._inner[0].mean_latent(num_latent)

    def forward(self, styles, input_is_latent=False, input_is_tensor=False, noise=None, randomize_noise=True, truncation=1,
                truncation_latent=None, inject_index=None, return_latents=False, conditions=None, cond_layers=None,
                cond_weights=None, cond_type='SFT'):
        if cond_layers is not None and conditions is not None and cond_type == 'NOISE':
            raise NotImplementedError('unknown mod_type NOISE')      # feature_modulation raises the same (stylegan2_arch.py:594)
        # reference quirk (stylegan2_arch.py:555,570): input_is_latent alone skips the MLP *and* the broadcast
        if input_is_latent and not input_is_tensor:
            input_is_tensor = True
        return self._inner[0](styles, return_latents=return_latents, inject_index=inject_index, truncation=truncation,
                              truncation_latent=truncation_latent, input_is_latent=input_is_latent,
                              input_is_tensor=input_is_tensor, noise=noise, randomize_noise=randomize_noise,
                              conditions=conditions, cond_layers=cond_layers, cond_type=cond_type)
