"""``ood_faceGAN_e4e`` — the drop-in top level of the path (reference
src/archs/OOD_faceGAN_e4e_arch.py:28-347): same constructor surface (the ``network_g`` block of
options/test/E4E_Face_test.yml), ``forward(x, **kw) -> (out, lats)``, side channels ``.aligns``,
``.delta_latent``, ``.avg_latent``, ``.generator.size``, ``random_gen``; plus the build-defined W+
refinement (``invert``) that the north star adds in front of the OOD forward.

The e4e encoder (the step *before* the path) is pluggable: ``self.encoder`` is any
callable ``encoder(x256, return_feats=True) -> (lats (B,18,512), feats[>=4])``; alternatively the
encoder outputs can be passed to ``forward`` as ``enc_lats=`` / ``enc_feats=``."""
import math
import os

import numpy as np
import torch
from torch import nn

from . import ops, samm
from .engine import WPlusInverter
from .modules import Generator
from .synth import generator_channels


class Registry:
    """name -> class registry, the plugin mechanism of BasicSR (basicsr/utils/registry.py:30-66)."""

    def __init__(self, name):
        self._name, self._obj_map = name, {}

    def register(self, obj=None):
        def deco(o):
            assert o.__name__ not in self._obj_map, f'{o.__name__} already registered in {self._name}'
            self._obj_map[o.__name__] = o
            return o
        return deco if obj is None else deco(obj)

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map

    def keys(self):
        return self._obj_map.keys()


ARCH_REGISTRY = Registry('arch')


def build_network(opt):
    """basicsr.archs.build_network (BasicSR/basicsr/archs/__init__.py:19-25): pops ``type``."""
    opt = dict(opt)
    return ARCH_REGISTRY.get(opt.pop('type'))(**opt)


class _MissingEncoder(nn.Module):
    channels = [64, 64, 128, 256, 512]
    progressive_stage = None

    def forward(self, x, return_feats=False):
        raise RuntimeError('no e4e encoder attached: pass enc_lats=/enc_feats= to forward(), or assign a callable '
                           'to model.encoder (the encoder is the step before the accelerated path, SURVEY.md §8f N1)')


@ARCH_REGISTRY.register()
class ood_faceGAN_e4e(nn.Module):
    def __init__(self,
                 out_size=1024, style_dim=512, n_mlp=8, channel_multiplier=2, narrow=1, merge='',
                 StyleGAN_pth=None, StyleGAN_pth_key='params_ema',
                 aug_alignment=False, aug_inputcolor=False,
                 stage='Inference', encoder='E4E', E4E_pth=None, avg_latent_pth=None,
                 optim_delta_latent=False, delta_latent_pth=None,
                 enable_modulation=True, modulation_type='NOISE', warp_scale=0.02,
                 blend_with_gen=True, ModSize=None,
                 progressiveModSize=[16, 32, 64, 128, 256], progressiveStart=20000,
                 progressiveStep=2000, progressiveStageSteps=[999999999], eval_path_length=None,
                 **kwargs):
        super().__init__()
        if aug_alignment or aug_inputcolor:
            # the reference reads undefined names for these options (OOD_faceGAN_e4e_arch.py:88-97 -> NameError)
            raise NotImplementedError('aug_alignment / aug_inputcolor are unusable in the reference as well')
        if encoder != 'E4E':
            raise NotImplementedError("only encoder='E4E' is on the accelerated path")
        if modulation_type != 'NOISE':
            raise NotImplementedError("only modulation_type='NOISE' (every shipped YAML)")
        if narrow != 1:
            raise NotImplementedError('narrow != 1')
        self.encoder_type = encoder
        log_outsize = int(math.log(out_size, 2))
        self.style_cnt = log_outsize * 2 - 2
        self.style_dim = style_dim
        self.channels = generator_channels(channel_multiplier, narrow)
        if kwargs.get('build_encoder', True):
            from .encoder import ProgressiveStage
            from .encoder_hip import Encoder4EditingHIP as Encoder4Editing     # the e4e encoder on the HIP kernels
            self.encoder = Encoder4Editing(num_layers=50, mode='ir_se', opts={'stylegan_size': out_size}, bn=True)
            self.encoder.progressive_stage = ProgressiveStage[stage]
        else:
            self.encoder = _MissingEncoder()   # encoder outputs are then passed to forward() as enc_lats / enc_feats
        self.aligns = {}
        self.log_outsize = int(math.log(256, 2))
        self.stage = stage
        if enable_modulation:
            self.feats_conv = nn.ModuleList()
            featsize = 256
            for i in range(4):
                self.feats_conv.append(nn.Conv2d(_MissingEncoder.channels[i], self.channels[featsize], 1, 1, 0))
                featsize //= 2
            self.modulation = nn.ModuleList()
            self.progressiveModSize = list(progressiveModSize)
            self.modulation_type = modulation_type
            self.blend_with_gen = blend_with_gen
            self.blend_cnt = kwargs.get('blend_cnt', 1)
            self.skip_SA = kwargs.get('skip_SA', False)
            self.randomTransform = None
            self.colorTransform = None
            self.ModSize = self.progressiveModSize.pop(0) if ModSize is None else ModSize
            self.warp_scale = warp_scale
            self.cycle_align = kwargs.get('cycle_align', 1)
            for i in range(self.log_outsize, 4, -1):
                chn = self.channels[2 ** i]
                chn_mul = 2 if modulation_type == 'SFT' else 1      # reference :110-112 (only matters with a mod_btn extractor)
                self.modulation.append(samm.StyledscaleNshfitBlock(chn, chn * chn_mul, style_dim, scale=warp_scale,
                                                                   btn=kwargs.get('mod_btn', None),
                                                                   cycle_align=self.cycle_align,
                                                                   diff_fAndg=kwargs.get('diff_fAndg', True)))
        else:
            self.modulation = None
            self.ModSize = 0
        self.generator = Generator(size=out_size, n_mlp=n_mlp, style_dim=style_dim, channel_multiplier=channel_multiplier)
        self.avg_latent = nn.Parameter(torch.zeros((1, style_dim)), requires_grad=False)
        if optim_delta_latent:
            self.delta_latent = nn.Parameter(torch.randn((1, self.style_cnt, style_dim)) * 0.1, requires_grad=True)
        else:
            self.delta_latent = nn.Parameter(torch.zeros((1, self.style_cnt, style_dim)), requires_grad=False)
        self.progressiveStageSteps = progressiveStageSteps
        if self.progressiveStageSteps is None:
            self.progressiveStageSteps = [progressiveStart + progressiveStep * i for i in range(self.style_cnt)]
        if StyleGAN_pth is not None:
            from .io import load_generator_checkpoint
            load_generator_checkpoint(self.generator, StyleGAN_pth, StyleGAN_pth_key)
        if E4E_pth is not None:
            from collections import OrderedDict
            enc_ckpt = torch.load(E4E_pth, map_location='cpu')
            enc_dict = OrderedDict((k[len('encoder.'):], v) for k, v in enc_ckpt['state_dict'].items() if k.startswith('encoder.'))
            self.encoder.load_state_dict(enc_dict, strict=True)
        if avg_latent_pth is not None:
            self.avg_latent.data = torch.load(avg_latent_pth, map_location='cpu')
        if delta_latent_pth is not None:
            self.delta_latent.data = torch.load(delta_latent_pth, map_location='cpu')
        self.eval_path_length = bool(eval_path_length) if eval_path_length is not None else False

    # ---------------------------------------------------------------- reference helpers
    def get_style_mlp(self, x):
        return self.generator.style(x)

    def random_gen(self, batch_size=1, gen=True):
        style = torch.randn((batch_size, self.style_dim), device=self.avg_latent.device)
        lats = self.get_style_mlp(style).unsqueeze(1).repeat(1, self.style_cnt, 1)
        out = self.generator(lats, input_is_tensor=True, input_is_latent=True)[0] if gen else None
        return out, lats

    def random_gen_center(self, scale=0.1, gen=True):
        lats = self.avg_latent + (torch.randn_like(self.avg_latent) * scale)
        lats = lats.unsqueeze(1).repeat(1, self.style_cnt, 1)
        out = self.generator(lats, input_is_tensor=True, input_is_latent=True)[0] if gen else None
        return out, lats

    def feats2condition(self, feats, **kwargs):
        conditions = []
        if self.ModSize > 0:
            max_size = int(np.floor(math.log(self.ModSize, 2)))
            min_size = int(np.floor(math.log(feats[-1].shape[-1], 2)))
            cond_len = min(max((1 + max_size - min_size), 0), len(feats))
            conditions = [[None, None] for _ in range(cond_len)]
        return conditions

    def _cond_hook(self, k, raw, style, noise, noise_weight):
        """feats2condition_callback (OOD_faceGAN_e4e_arch.py:224-242) in its algebraically reduced
        form: the layer becomes aligned_target + w*noise (model.py:292), so the aligned feature
        itself is returned and the engine adds noise/bias/activation."""
        ind = k + 1
        feat = self.feats[-ind]
        mod = self.modulation[-ind]
        aligned = self.aligns[ind - 1] if ind > 1 else None
        cond, align = mod(feat, style, image=raw, aligned=aligned)
        self.aligns[ind] = align
        return cond

    def encode(self, x, **kwargs):
        """Step 1-2 of forward (:256-267): encoder at 256², + avg_latent + delta_latent (+ truncation)."""
        enc_lats, enc_feats = kwargs.get('enc_lats', None), kwargs.get('enc_feats', None)
        if enc_lats is None or (self.modulation is not None and enc_feats is None):
            x256 = samm.resize_bilinear(x, 256)
            with torch.no_grad():
                if self.encoder.training:       # the reference calls eval() on every forward (:257): 1 ms of host time per call over 529 modules
                    self.encoder.eval()
                enc_lats, enc_feats = self.encoder(x256, return_feats=True)
        lats = enc_lats + self.avg_latent.reshape(1, 1, -1) + self.delta_latent
        truncation = kwargs.get('truncation', 1.0)
        if truncation < 1.0:
            lats = self.avg_latent.reshape(1, 1, -1) * (1. - truncation) + (lats * truncation)
        return lats.contiguous(), enc_feats

    def forward(self, x, **kwargs):
        if kwargs.get('random_gen', False):
            return self.random_gen(batch_size=kwargs.get('batch_size', 1), gen=kwargs.get('gen', True))
        lats, enc_feats = self.encode(x, **kwargs)
        if kwargs.get('lats', None) is not None:                 # W+ refined latents replace the encoder's
            lats = kwargs['lats']
        return self._ood_forward(x, lats, enc_feats, **{k: v for k, v in kwargs.items() if k not in ('lats', 'enc_lats', 'enc_feats')})

    def _ood_forward(self, x, lats, enc_feats, **kwargs):
        """generate() of the reference (e4e :268-313 / restyle :246-283): feats_conv -> hooked generator -> mask blend."""
        self.ori_lats = lats
        noise = kwargs.get('noise', None)
        if self.modulation is None:
            out, _ = self.generator(lats, input_is_tensor=True, input_is_latent=True, noise=noise)
            return out, lats
        self.feats = [samm.conv1x1(enc_feats[i], self.feats_conv[i].weight, self.feats_conv[i].bias) for i in range(4)]
        self.lats = lats
        self.aligns = {}
        conditions = self.feats2condition(self.feats)
        cond_ind = [(2 * (k + 2)) + 1 for k in range(len(conditions))]
        out, _ = self.generator(lats, input_is_tensor=True, input_is_latent=True, conditions=conditions,
                                cond_layers=cond_ind, cond_type=self.modulation_type, cond_hook=self._cond_hook,
                                noise=noise)
        if self.blend_with_gen:
            if self.skip_SA:
                out, _ = self.generator(lats, input_is_tensor=True, input_is_latent=True, noise=noise)
            for _ in range(self.blend_cnt):
                out = self.blend(x, out, alpha_scale=None)
        return out, lats

    def blending_mask(self):
        """:315-339 — compose the up-sampled alpha channels (coarse -> fine), clip; stores aligns[size]."""
        self.aligns.pop(self.generator.size, None)
        keys = sorted(self.aligns.keys())
        if not keys:
            return None
        alpha, _ = samm.mask_blend([self.aligns[k] for k in keys], size=self.generator.size)
        self.aligns[self.generator.size] = alpha.repeat(1, 3, 1, 1)
        return alpha

    def blend(self, target, output, detach=True, alpha_scale=None):
        """:341-347; with alpha_scale=None the mask is composed and applied in one fused kernel."""
        if alpha_scale is not None:
            return alpha_scale * target + output * (1 - alpha_scale)
        self.aligns.pop(self.generator.size, None)
        keys = sorted(self.aligns.keys())
        if not keys:
            return None
        alpha, out = samm.mask_blend([self.aligns[k] for k in keys], target, output, size=self.generator.size)
        self.aligns[self.generator.size] = alpha.repeat(1, 3, 1, 1)
        return out

    # ---------------------------------------------------------------- build-defined: W+ refinement
    def invert(self, x, steps=100, lr=0.01, noise=None, streams=1, use_graph=False, lpips_weight=0.0, lpips_state=None, **kwargs):
        """Optimisation-based inversion: w0 = encoder latents (+avg+delta), ``steps``
        Adam steps on per-image MSE with fixed noise, then ONE full OOD forward with the refined
        latents (masks + blend).  Returns (out, lats, losses[steps,B]).  ``lpips_weight`` > 0 adds that multiple of LPIPS(alex) per image to the
        loss (opt-in; ``losses`` is then the total, ``self.last_loss_terms`` holds the two tables).
        ``streams`` (opt-in; ``bench.py`` uses 2, the CLI's ``inversion.streams`` sets it): the W+ loop advances the batch as
        that many independent sub-batches on concurrent HIP streams (images are independent; the HBM-bound layout
        kernels of one sub-batch run beside the matrix kernels of the other: +4 % at batch 8).  The
        default 1 keeps every kernel alone on the GPU: concurrent queues are only exact for kernels built without
        packed-fp32 instructions (this library's are; a caller's own torch-ROCm work on another stream is not covered).
        Results are bit-reproducible run to run for a given (batch, streams); they are NOT bit-invariant under the
        stream count or an image's position in the batch — kernel selection follows the sub-batch geometry (8-wave
        kernels from 128 work items) and every sub-batch carries its own range scales — the difference is fp32
        rounding on the first step (loss within 1e-5 relative) and Adam's sign choice for ~0 gradient coordinates afterwards
        (0.003 % of the coordinates, losses within 1e-3 relative after three steps: ``tests/test_hip_generator.py``)."""
        lats0, enc_feats = self.encode(x, **kwargs)
        B = x.shape[0]
        if noise is None:
            noise = [n.expand(B, -1, -1, -1).contiguous() for n in self.generator.make_noise()]
        lp = None
        if lpips_weight:
            # loss = MSE + lpips_weight * LPIPS(alex) (reference loss class: src/losses/lpips_loss.py:13-34, min_max = the generator's (-1, 1)).
            # ``lpips_state``: the lpips package's state dict; None = the seeded stand-in (weights absent here: parity unpinned)
            from .lpips import LPIPSAlex
            from .synth import lpips_state as _seeded
            key = id(lpips_state)
            if getattr(self, '_lpips_key', None) != key:
                st = lpips_state if lpips_state is not None else _seeded(0)
                self._lpips_net, self._lpips_key = LPIPSAlex({k: v.to(x.device) for k, v in st.items()}, min_max=(-1.0, 1.0)), key
            lp = self._lpips_net
        inv = WPlusInverter(self.generator.engine(), lr=lr, lpips=lp, lpips_weight=lpips_weight)
        w, losses = inv.invert(x, lats0, noise, steps=steps, streams=streams, use_graph=use_graph)
        self.last_loss_terms, self.last_invert_stats, self.last_invert_plan = inv.last_terms, inv.last_stats, inv.last_plan
        kw = {k: v for k, v in kwargs.items() if k not in ('noise_passes', 'truncation', 'enc_lats', 'enc_feats', 'lats', 'noise')}
        out, lats = self._ood_forward(x, w, enc_feats, noise=noise, **kw)
        # An inversion returns finished results.  Without this the host runs ahead into the next inversion's set-up while the device still works
        # on this one: blocks this inversion's side streams freed are then not yet reusable, the caching allocator goes to the driver for new ones
        # and every inversion pays ~20 ms (measured, bench.py back-to-back inversions on one box: 1054-1061 ms without, 1036-1047 ms with it —
        # profiles/r6_invert_sync_ab.txt; until round 6 a host read of the
        # carried-scale flag in Generator.forward happened to provide this synchronisation)
        torch.cuda.current_stream().synchronize()
        return out, lats, losses


class GraphedForward:
    """``model(x)`` replayed from a captured hipGraph — for the single-image latency the reference's CLI reports ("Average process
    time", run_ood_faceGAN_inversion.py:167-172,187): at B = 1 the forward is ~800 launches of 5-250 us and the host call overhead and
    the gaps between dependent launches are ~6 % of the 15 ms (at B = 8 nothing).  One graph per input shape, captured on first use
    after two eager warm-up calls; ``noise=`` (the 17 generator maps) is copied into static buffers, otherwise every replay draws
    fresh noise (PyTorch registers its generator with the graph).  Returned tensors and ``model.aligns`` are the graph's static
    buffers: they are overwritten by the next call.  Call ``reset()`` after changing any weight (packed copies are part of the graph).
    Works for the three variants (their forwards make no host-side decisions on tensor values)."""

    def __init__(self, model):
        self.model, self._cache, self._side = model, {}, None

    def reset(self):
        """Drop the captured graphs (and, with them, their private pools); the scratch buffers the library-side pools keep per
        stream handle (ops.sform_scratch / conv_workspace) are released too — they were created for this object's side stream."""
        self._cache = {}
        if self._side is not None:
            from . import ops
            ops.drop_stream_scratch(self._side.cuda_stream)

    @torch.no_grad()
    def __call__(self, x, noise=None):
        key = (tuple(x.shape), noise is not None)
        ent = self._cache.get(key)
        if ent is None:
            sx = x.clone()
            sn = None if noise is None else [n.clone() for n in noise]
            # ONE side stream per object, used for the warm-up AND the capture: the S-form scratch buffers and conv workspaces are
            # keyed by the stream handle (ops.sform_scratch), so the buffers the warm-up creates are the ones the captured launches
            # use — nothing is allocated (or zero-filled) inside the graph, and nothing is left behind under another handle
            if self._side is None:
                self._side = torch.cuda.Stream(device=x.device)
            side = self._side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                  # packs weights, fills the memo caches, grows the allocator pools
                    self.model(sx, noise=sn) if sn is not None else self.model(sx)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                out, lats = self.model(sx, noise=sn) if sn is not None else self.model(sx)
            gen = getattr(self.model, 'generator', None)
            eng = getattr(gen, '_engine_obj', None) if gen is not None else None
            # the range state THIS graph's launches read and write (one per batch size, engine._range): the check after a replay must look
            # at this object, not at whatever batch size ran last
            ent = self._cache[key] = dict(graph=graph, x=sx, noise=sn, out=out, lats=lats, aligns=dict(self.model.aligns),
                                          rng=getattr(eng, 'fwd_range', None))
        ent['x'].copy_(x)
        if noise is not None:
            for d, n in zip(ent['noise'], noise):
                d.copy_(n)
        ent['graph'].replay()
        # the captured forward runs on carried range scales (modules.Generator.forward) and cannot read their check flag inside the graph:
        # read it here (one small device-to-host copy) and repeat the call eagerly with measured scales if it is set
        gen = getattr(self.model, 'generator', None)
        eng = gen._engine_obj if gen is not None and getattr(gen, '_engine_obj', None) is not None else None
        rng = ent.get('rng')
        if eng is not None and rng is not None and rng.violated():
            eng.reset_fwd_state(rng.B)
            out, lats = self.model(x, noise=noise) if noise is not None else self.model(x)
            return out, lats
        self.model.aligns = dict(ent['aligns'])
        return ent['out'], ent['lats']


@ARCH_REGISTRY.register()
class ood_faceGAN_restyle(ood_faceGAN_e4e):
    """The ReStyle variant (reference src/archs/OOD_faceGAN_restyle_arch.py:29-375): the latent code is
    predicted iteratively — ``enc_cycle`` passes of a 6-channel encoder over [pool256(x), pool256(current reconstruction)],
    starting from the average image — and the OOD forward (SAMM hooks, mask, blend) then runs exactly as in
    ``ood_faceGAN_e4e``.  Same constructor surface, ``forward(x, **kw) -> (out, lats)`` and side channels.

    ``ReStyle_pth``: a checkpoint with ``state_dict`` (``encoder.*`` keys), ``latent_avg`` (style_cnt, style_dim) and
    ``opts`` (``encoder_type``, ``input_nc``), as the reference loads it (:67-83).  Only ``ProgressiveBackboneEncoder``
    is supported (the ResNet-34 variant needs torchvision).

    The reference draws fresh noise in every generator pass (it never forwards ``noise=``).  For reproducible runs
    ``forward`` takes ``noise_passes=[avg-image pass (batch 1), cycle pass 1, ..., final pass]``, each a list of the 17
    per-layer maps; ``noise=`` sets the final pass only."""

    def __init__(self, out_size=1024, style_dim=512, encoder='ReStyle', ReStyle_pth=None, enc_cycle=2, avg_latent_pth=None, **kwargs):
        if encoder != 'ReStyle':
            raise NotImplementedError("ood_faceGAN_restyle: encoder must be 'ReStyle'")
        if ReStyle_pth is None:
            raise AssertionError('ReStyle_pth is required (reference :66)')
        stage = kwargs.pop('stage', 'Inference')
        kwargs.pop('build_encoder', None)
        kwargs.pop('E4E_pth', None)
        super().__init__(out_size=out_size, style_dim=style_dim, encoder='E4E', stage=stage, build_encoder=False, **kwargs)
        self.encoder_type = encoder
        from collections import OrderedDict
        from .encoder import ProgressiveStage
        enc_ckpt = torch.load(ReStyle_pth, map_location='cpu') if isinstance(ReStyle_pth, (str, bytes)) or hasattr(ReStyle_pth, '__fspath__') else ReStyle_pth
        opts = dict(enc_ckpt['opts'])
        if opts.get('encoder_type') != 'ProgressiveBackboneEncoder':
            raise NotImplementedError(f"ReStyle encoder_type {opts.get('encoder_type')!r}: only ProgressiveBackboneEncoder")
        from .encoder_hip import ProgressiveBackboneEncoderHIP as Enc
        self.encoder = Enc(num_layers=50, mode='ir_se', n_styles=self.style_cnt, opts=opts)
        enc_dict = OrderedDict((k[len('encoder.'):], v) for k, v in enc_ckpt['state_dict'].items() if k.startswith('encoder.'))
        self.encoder.load_state_dict(enc_dict, strict=True)
        self.encoder.progressive_stage = ProgressiveStage[stage]
        self.avg_latent = nn.Parameter(enc_ckpt['latent_avg'].detach().clone().float().reshape(self.style_cnt, style_dim), requires_grad=False)
        self.enc_cycle = enc_cycle
        self.avg_img = None

    def face_pool(self, x):
        return samm.avgpool(x, 256)

    def random_gen_center(self, scale=0.1, gen=True, noise=None):
        lats = (self.avg_latent + (torch.randn_like(self.avg_latent) * scale)).unsqueeze(0)
        out = self.generator(lats, input_is_tensor=True, input_is_latent=True, noise=noise)[0] if gen else None
        return out, lats

    def encode(self, x, **kwargs):
        """:288-318 — iterative latent prediction; returns (lats, encoder feats of the last cycle)."""
        passes = list(kwargs.get('noise_passes', None) or [])
        take = lambda: passes.pop(0) if passes else None          # noqa: E731
        with torch.no_grad():
            if self.avg_img is None:
                avg_img, _ = self.random_gen_center(scale=0, noise=take())
                self.avg_img = self.face_pool(avg_img)
            elif len(passes) > self.enc_cycle:                    # avg image cached: its noise entry is not needed
                passes.pop(0)
            if self.encoder.training:
                self.encoder.eval()
            x256 = self.face_pool(x)
            lats, feats = self.encoder(torch.cat([x256, self.avg_img.repeat(x.shape[0], 1, 1, 1)], dim=1), return_feats=True)
            lats = lats + self.avg_latent.unsqueeze(0)
            for _ in range(self.enc_cycle - 1):
                # both reference branches (:308-311) end in the plain, hook-free generator pass
                new_x = self.generator(lats.contiguous(), input_is_tensor=True, input_is_latent=True, noise=take())[0]
                delta, feats = self.encoder(torch.cat([x256, self.face_pool(new_x)], dim=1), return_feats=True)
                lats = lats + delta
        lats = lats + self.delta_latent
        truncation = kwargs.get('truncation', 1.0)
        if truncation < 1.0:
            lats = self.avg_latent.unsqueeze(0) * (1. - truncation) + (lats * truncation)
        self._final_noise = take()
        return lats.contiguous(), feats

    def forward(self, x, **kwargs):
        if kwargs.get('random_gen', False):
            return self.random_gen(batch_size=kwargs.get('batch_size', 1), gen=kwargs.get('gen', True))
        if kwargs.get('enc_lats', None) is not None:
            raise NotImplementedError('enc_lats= is the e4e entry; the ReStyle encoder is iterative')
        lats, feats = self.encode(x, **kwargs)
        kw = {k: v for k, v in kwargs.items() if k not in ('noise_passes', 'truncation')}
        if kw.get('noise', None) is None:
            kw['noise'] = self._final_noise
        return self._ood_forward(x, lats, feats, **kw)


@ARCH_REGISTRY.register()
class ood_faceGAN_FeatureStyle(ood_faceGAN_e4e):
    """The Feature-Style variant (reference src/archs/OOD_faceGAN_featureStyle_arch.py:28-334): latents from
    ``fs_encoder_v2`` (IResNet-50 trunk, pooled descriptors, 18 linear heads) on the 256² average-pooled input, SAMM taps
    = stem + first three stages, then the shared OOD forward.  ``FeatureStyle_pth`` is the encoder's state dict (loaded
    strictly, :74-79).  ``arcface_model_path`` is accepted and not read: the reference only uses it to initialise the trunk
    before ``FeatureStyle_pth`` overwrites every parameter.  Like the reference (:277-291) the encoder's ``content`` feature
    is computed and NOT injected — ``generate(lats, feats, x)`` leaves ``contents=None``."""

    def __init__(self, out_size=1024, style_dim=512, StyleGAN_pth_key='g_ema', encoder='FeatureStyle', FeatureStyle_pth=None,
                 arcface_model_path=None, avg_latent_pth=None, **kwargs):
        if encoder != 'FeatureStyle':
            raise NotImplementedError("ood_faceGAN_FeatureStyle: encoder must be 'FeatureStyle'")
        if FeatureStyle_pth is None:
            raise AssertionError('FeatureStyle_pth is required (reference :73)')
        kwargs.pop('build_encoder', None)
        kwargs.pop('E4E_pth', None)
        super().__init__(out_size=out_size, style_dim=style_dim, StyleGAN_pth_key=StyleGAN_pth_key, encoder='E4E', build_encoder=False, **kwargs)
        self.encoder_type = encoder
        from .encoder_hip import fs_encoder_v2HIP as Enc
        self.encoder = Enc(n_styles=self.style_cnt, stride=(2, 2))
        enc_ckpt = torch.load(FeatureStyle_pth, map_location='cpu') if isinstance(FeatureStyle_pth, (str, bytes)) or hasattr(FeatureStyle_pth, '__fspath__') else FeatureStyle_pth
        self.encoder.load_state_dict(enc_ckpt, strict=True)
        self.avg_latent = nn.Parameter(torch.zeros((self.style_cnt, style_dim)), requires_grad=False)
        if avg_latent_pth is not None:
            self.avg_latent.data = torch.load(avg_latent_pth, map_location='cpu')

    def face_pool(self, x):
        return samm.avgpool(x, 256)

    def encode(self, x, **kwargs):
        """:268-285"""
        with torch.no_grad():
            if self.encoder.training:
                self.encoder.eval()
            lats, self.content, feats = self.encoder(self.face_pool(x), return_feats=True)
        lats = lats + self.avg_latent.unsqueeze(0) + self.delta_latent
        truncation = kwargs.get('truncation', 1.0)
        if truncation < 1.0:
            lats = self.avg_latent.unsqueeze(0) * (1. - truncation) + (lats * truncation)
        return lats.contiguous(), feats

    def forward(self, x, **kwargs):
        if kwargs.get('random_gen', False):
            return self.random_gen(batch_size=kwargs.get('batch_size', 1), gen=kwargs.get('gen', True))
        if kwargs.get('enc_lats', None) is not None:
            raise NotImplementedError('enc_lats= is the e4e entry')
        lats, feats = self.encode(x, **kwargs)
        return self._ood_forward(x, lats, feats, **{k: v for k, v in kwargs.items() if k != 'truncation'})
