"""Generator schedule on the HIP ops: forward (with optional SAMM hooks), backward w.r.t. the W+
latents, and the W+ Adam inversion loop.

``GeneratorEngine`` owns the *prepared* form of the frozen generator weights:
  * every 3x3 weight packed K-major for the MFMA implicit-GEMM kernels (forward and input-gradient
    layouts), scaled by 1/sqrt(fan_in) once;
  * per-(co,ci) squared tap sums for the factorised demodulation;
  * all 26 modulation matrices concatenated (rows grouped by latent index) so that the style affine
    of a whole forward is one MFMA contraction and its backward one kernel.
Layer order / latent indexing follow reference src/ops/StyleGAN/model.py:548-576."""
import math
import re

import os

import torch

from . import ops
from ._lib import ACT_LRELU, ACT_NONE, CONV_S1, CONV_S2, CONV_T2
from .synth import generator_channels, make_kernel


_Cols = ops.Cols


class _Layer:
    __slots__ = ('name', 'kind', 'cin', 'cout', 'res', 'lat', 'row', 'drow', 'wpk', 'wpk_bwd', 'wsq', 'w_rgb',
                 'bias', 'noise_w', 'scale', 'noise_idx', 'src', 'sidx', 'wpk_vb')


def _pad_channels_to_16(state, prefix, log_size, ch):
    """The matrix kernels take 16-channel blocks.  A generator with other channel counts (``StyleGAN2Generator(narrow=...)``, stylegan2_arch.py:422,
    435-443: 8 channels at 1024² for narrow = 0.25) runs as the SAME function on zero-padded tensors: padded input channels get zero conv / ToRGB
    weights and a zero modulation row and bias (style 0), padded output channels zero weights and bias — they only ever hold lrelu(noise), which
    every reader multiplies by zero, and no gradient reaches the latents through them.  ToRGB weights of a padded layer carry sqrt(padded / real):
    the ops compute 1/sqrt(C) from the count they see.  Returns (state with the padded tensors, {res: padded count})."""
    up16 = lambda c: (c + 15) // 16 * 16
    chp = {r: up16(c) for r, c in ch.items()}
    out = dict(state)

    def pad(key, sizes):            # sizes: {dim: new size}
        t = state[prefix + key]
        shape = list(t.shape)
        for d, n in sizes.items():
            shape[d] = n
        if shape == list(t.shape):
            return
        new = torch.zeros(shape, dtype=t.dtype, device=t.device)
        new[tuple(slice(0, n) for n in t.shape)] = t
        out[prefix + key] = new

    def styled(name, cin, cout):
        pad(f'{name}.conv.weight', {1: up16(cout), 2: up16(cin)})
        pad(f'{name}.conv.modulation.weight', {0: up16(cin)})
        pad(f'{name}.conv.modulation.bias', {0: up16(cin)})
        pad(f'{name}.activate.bias', {0: up16(cout)})

    def rgb(name, cin):
        pad(f'{name}.conv.weight', {2: up16(cin)})
        # every ToRGB op derives its fan-in scale 1/sqrt(C) from the channel count it is handed (the padded one): fold the ratio into the weights
        if up16(cin) != cin:
            out[prefix + f'{name}.conv.weight'] = out[prefix + f'{name}.conv.weight'] * math.sqrt(up16(cin) / cin)
        pad(f'{name}.conv.modulation.weight', {0: up16(cin)})
        pad(f'{name}.conv.modulation.bias', {0: up16(cin)})

    pad('input.input', {1: up16(ch[4])})
    styled('conv1', ch[4], ch[4])
    rgb('to_rgb1', ch[4])
    cin = ch[4]
    for j in range(log_size - 2):
        cout = ch[2 ** (j + 3)]
        styled(f'convs.{2 * j}', cin, cout)
        styled(f'convs.{2 * j + 1}', cout, cout)
        rgb(f'to_rgbs.{j}', cout)
        cin = cout
    return out, chp


class GeneratorEngine:
    def __init__(self, state, size, style_dim=512, channel_multiplier=2, prefix='', with_backward=True, precision=None, narrow=1,
                 blur_kernel=(1, 3, 3, 1), upsample_kernel=(1, 3, 3, 1)):
        self.size, self.style_dim = size, style_dim
        self.precision = precision or ops.PRECISION
        # 'f16s-g2' (round 6): the split-f16 arithmetic with the BACK-PROPAGATED gradient rounded to f16 before each contraction —
        # g_hi * (w_hi + w_lo), two matrix instructions per product in the input-gradient convs; the forward is unchanged
        self.grad_hi_only = self.precision == 'f16s-g2'
        if self.grad_hi_only:
            self.precision = 'f16s'
        self.sform = (self.precision == 'f16s') and ops.USE_SFORM
        self.log_size = int(math.log2(size))
        self.n_latent = self.log_size * 2 - 2
        self.num_layers = (self.log_size - 2) * 2 + 1
        ch = generator_channels(channel_multiplier, narrow)
        used = [ch[2 ** i] for i in range(2, self.log_size + 1)]
        if any(c < 1 for c in used):
            raise ValueError(f'channel counts {used} (narrow={narrow}, channel_multiplier={channel_multiplier})')
        self.padded = any(c % 16 for c in used)
        self.real_channels = dict(ch)
        if self.padded:
            state, _ = _pad_channels_to_16(state, prefix, self.log_size, ch)
        g = lambda k: state[prefix + k].detach().float().contiguous()
        dev = g('input.input').device
        if dev.type != 'cuda':
            raise RuntimeError('GeneratorEngine needs its parameters on a ROCm device (no CPU fallback)')
        self.device = dev
        self.const_input = g('input.input')
        # The kernels of Blur(upsample_factor=2) in the up-convs (model.py:72-81,199-205) and of Upsample in ToRGB's skip path (model.py:30-48,
        # 353-372).  Both are registered buffers: what the state holds is what the reference runs (a loaded checkpoint overrides the constructor's
        # taps — and model.py:455 builds ToRGB WITHOUT the constructor's ``blur_kernel``, so the two families may differ); the keyword taps serve
        # states without these keys (BasicSR layout: stylegan2_arch.py:61,115 keeps them as plain attributes)
        def family(pattern, taps, what):
            pat = re.compile('^' + re.escape(prefix) + pattern + r'\.kernel$')
            held = [v.detach().float().cpu() for k, v in state.items() if pat.match(k)]
            k4 = held[0] if held else make_kernel(tuple(taps)) * 4.0
            if tuple(k4.shape) != (4, 4):
                raise NotImplementedError(f'{what} kernel of shape {tuple(k4.shape)}: the fused producers are built for four taps (every shipped config: [1,3,3,1])')
            if any(h.shape != k4.shape or not torch.equal(h, k4) for h in held):
                raise NotImplementedError(f'the {what} kernels of the state differ between layers')
            return k4.contiguous()

        k4 = family(r'convs\.\d+\.conv\.blur', blur_kernel, 'Blur')
        self.k4x4 = k4.to(dev)
        self.k4x4_flip = torch.flip(self.k4x4, [0, 1]).contiguous()
        self.k4x4_rank1 = bool(torch.linalg.matrix_rank(k4.double()) == 1)
        self.k_up = family(r'to_rgbs\.\d+\.upsample', upsample_kernel, 'Upsample').to(dev)
        self.k_up_flip = torch.flip(self.k_up, [0, 1]).contiguous()
        layers = []

        up16 = lambda c: (c + 15) // 16 * 16

        def styled(name, cin, cout, res, lat, up, nidx):        # cin / cout: the REAL counts (fan-in scale); the layer runs on the padded ones
            L = _Layer()
            L.name, L.kind, L.cin, L.cout, L.res, L.lat, L.noise_idx = name, ('up' if up else 'conv'), up16(cin), up16(cout), res, lat, nidx
            w = g(f'{name}.conv.weight')[0]                 # (Co,Ci,3,3)
            L.scale = 1.0 / math.sqrt(cin * 9)
            L.wpk = ops.pack_conv3x3(w, L.scale, transpose=False, flip=False, precision=self.precision)
            L.wpk_bwd = ops.pack_conv3x3(w, L.scale, transpose=True, flip=not up, precision=self.precision) if with_backward else None
            if L.wpk_bwd is not None:
                L.wpk_bwd.x_hi_only = self.grad_hi_only
            L.wsq = ops.weight_sqsum(w)
            L.wpk_vb = None
            L.bias = g(f'{name}.activate.bias')
            L.noise_w = g(f'{name}.noise.weight')
            layers.append(L)
            return L

        def rgb(name, cin, res, lat):
            L = _Layer()
            L.name, L.kind, L.cin, L.cout, L.res, L.lat = name, 'rgb', up16(cin), 3, res, lat
            L.w_rgb = g(f'{name}.conv.weight').reshape(3, up16(cin)).contiguous()
            L.bias = g(f'{name}.bias').reshape(3).contiguous()
            L.scale = 1.0 / math.sqrt(up16(cin))    # with the weights of a padded layer x sqrt(padded / real) (_pad_channels_to_16)
            layers.append(L)
            return L

        styled('conv1', ch[4], ch[4], 4, 0, False, 0)
        rgb('to_rgb1', ch[4], 4, 1)
        cin, i = ch[4], 1
        for j in range(self.log_size - 2):
            res = 2 ** (j + 3)
            cout = ch[res]
            styled(f'convs.{2 * j}', cin, cout, res, i, True, 2 * j + 1)
            styled(f'convs.{2 * j + 1}', cout, cout, res, i + 1, False, 2 * j + 2)
            rgb(f'to_rgbs.{j}', cout, res, i + 2)
            cin, i = cout, i + 2
        self.layers = layers
        # fused backward producers (csrc/bwd_producers.hip): per-layer range scale carried from one W+ step to the next;
        # the flags below exist for the exact-scale fallback and for A/B tests (tests flip them on the instance)
        self.fused_bwd = True
        self.bwd_state, self.bwd_flag = {}, None
        styled = [L for L in layers if L.kind != 'rgb']
        self.layers_styled_last = styled[-1]
        self.fused_fwd = True
        self.next_conv = {a.name: b for a, b in zip(styled[:-1], styled[1:]) if a.kind == 'up' and b.kind == 'conv'}
        # ToRGB layer -> the up-sampling conv that reads the same feature map next (its S-form input is written by ToRGB)
        self.rgb_next_up = {a.name: b for a, b in zip(layers[:-1], layers[1:]) if a.kind == 'rgb' and b.kind == 'up'}
        self.conv_next_rgb = {a.name: b for a, b in zip(layers[:-1], layers[1:]) if a.kind == 'conv' and b.kind == 'rgb'}
        self.by_name = {L.name: L for L in layers}
        self.fuse_act_bwd = True     # activation backward of the conv layers inside the stride-2 conv's epilogue (carried scales)
        self.hi_records = os.environ.get('OODGAN_HI_RECORDS', '1') != '0'    # f16s-g2: 32-byte hi-only gradient records between the blur^T producer and the stride-2 conv
        self.fused_rgb = True
        self.fuse_x = True           # 1024² level: F-form activations, the strip convs convert their input themselves (conv_f16s_stripx.hip)
        # the up-conv of that level in ONE pass (transposed conv + blur + noise + bias + activation -> F-form, csrc/conv_f16s_upvb.hip):
        # the blur's vertical pass folded into two 3x3 weight sets — prepared here, once, for the layer in front of the last conv
        self.fuse_up = True
        self.save_sform_only = os.environ.get('OODGAN_SAVE_SFORM_ONLY', '1') != '0'    # W+ loop: those layers' saved activation only as that S-form
        self.fuse_conv_rgb_64 = True     # ... from the 64² level on (the separate ToRGB pass started at 128²)
        self.sform_only_max_res = None   # A/B: largest resolution whose conv activation is saved only as the S-form (None: all eligible levels)
        self.fuse_conv_rgb = os.environ.get('OODGAN_FUSE_CONV_RGB', '1') != '0'    # ToRGB sums + the next up-conv's S-form from the 8-wave conv's epilogue (128² ... 512² levels, carried scales)
        self.plain_one_pass = os.environ.get('OODGAN_PLAIN_ONE_PASS', '1') != '0'     # the plain forward's last level through the one-pass up-conv + in-kernel conversion too (A/B flag)
        Lup = next((a for a, b in zip(styled[:-1], styled[1:]) if a.kind == 'up' and b is styled[-1]), None)
        if (Lup is not None and self.precision == 'f16s' and self.sform and with_backward and Lup.cout % 32 == 0 and Lup.cin % 16 == 0
                and Lup.cout <= 32 and size >= 64):
            Lup.wpk_vb = ops.pack_upconv_vblur(g(f'{Lup.name}.conv.weight')[0], Lup.scale, self.k4x4)
        ops.xf_supported(1, 32, 32, 8, 32)      # first call initialises that path (allocations must not fall into a stream capture)
        self.batched_tail = True
        src = 'input'
        for L in layers:            # producer of every layer's input feature
            L.src = src
            if L.kind != 'rgb':
                src = L.name
        # concatenated modulation matrix, rows grouped by latent index (execution order already is)
        rows, drows, wl, bl, rl = 0, 0, [], [], []
        lat_start = [0] * (self.n_latent + 1)
        row_layer, drow_layer, sidx = [], [], 0
        for L in layers:
            L.row = rows
            wl.append(g(f'{L.name}.conv.modulation.weight'))
            bl.append(g(f'{L.name}.conv.modulation.bias'))
            rl += [L.lat] * L.cin
            rows += L.cin
            lat_start[L.lat + 1] = rows
            if L.kind != 'rgb':
                L.drow = drows
                drows += L.cout
                L.sidx = sidx                       # index of the styled conv: row of the forward range-scale table
                row_layer += [sidx] * L.cin
                drow_layer += [sidx] * L.cout
                sidx += 1
            else:
                L.sidx = -1
                row_layer += [-1] * L.cin           # ToRGB colour weights use the true style
        for l in range(1, self.n_latent + 1):
            lat_start[l] = max(lat_start[l], lat_start[l - 1])
        self.R, self.DR = rows, drows
        self.wcat = torch.cat(wl, 0).contiguous()
        self.bcat = torch.cat(bl, 0).contiguous()
        self.row_lat = torch.tensor(rl, dtype=torch.int32, device=dev)
        self.lat_start = torch.tensor(lat_start, dtype=torch.int32, device=dev)
        self.row_layer = torch.tensor(row_layer, dtype=torch.int32, device=dev)
        self.drow_layer = torch.tensor(drow_layer, dtype=torch.int32, device=dev)
        self.n_styled = sidx
        # forward range control of the split-f16 path (ops.FwdRange): per-step state, created for the batch size in use
        self.fwd_range = None
        self.carry_range = True      # False: every forward measures its scales exactly (the fallback after a violation)
        self.stored_noises = [g(f'noises.noise_{k}') for k in range(self.num_layers)] if (prefix + 'noises.noise_0') in state else None
        self.saved = None

    def clone_shared(self):
        """A second engine over the SAME prepared weights (read-only) with its own per-step state."""
        import copy
        e = copy.copy(self)
        e.saved = None
        e.bwd_state, e.bwd_flag = {}, None
        e.fwd_range = None
        e._ranges = {}
        return e

    def reset_fwd_state(self, B=None):
        """Forget the carried forward range scales (new images / new batch): the next forward measures them exactly.
        Every batch size's state is reset (``B``: only that one) — a state left `valid` with another image set's scales would
        make the first forward at that batch size run in carry mode and the result depend on the call history."""
        ranges = self.__dict__.get('_ranges', {})
        todo = list(ranges.values()) if B is None else [r for r in (ranges.get(B),) if r is not None]
        if self.fwd_range is not None and (B is None or self.fwd_range.B == B) and not any(r is self.fwd_range for r in todo):
            todo.append(self.fwd_range)
        for r in todo:
            r.valid = False
            r.flag.zero_()
            r.vm.zero_()       # maxima an aborted carry pass left behind must not enter the next measurement

    def fwd_range_violated(self):
        """True if a carried forward scale left the exact window in any forward since reset_fwd_state() (host sync)."""
        return self.fwd_range is not None and self.fwd_range.violated()

    def _range(self, B):
        if self.fwd_range is None or self.fwd_range.B != B:
            # one state per batch size (an inversion alternates between the sub-batch of the W+ loop and the full batch of the OOD forward:
            # the latter's carried scales survive the former)
            ranges = self.__dict__.setdefault('_ranges', {})
            r = ranges.get(B)
            if r is None:
                # never evicted: captured hipGraphs (arch.GraphedForward, the W+ step graphs) hold raw pointers to a state's
                # q / vm / flag / s_sc / d_sc — a state is ~100 KB per batch size
                r = ranges[B] = ops.FwdRange(self.n_styled, B, self.R, self.DR, self.row_layer, self.drow_layer, self.device)
            self.fwd_range = r
        return self.fwd_range

    def reset_bwd_state(self):
        """Forget the carried range scales (new images / new batch): the next backward measures them exactly."""
        self.bwd_state = {}
        if self.bwd_flag is not None:
            self.bwd_flag.zero_()

    def bwd_scale_violated(self):
        """True if a carried range scale left the exact window in any step since reset_bwd_state() (host sync)."""
        return self.bwd_flag is not None and int(self.bwd_flag.item()) != 0

    # ------------------------------------------------------------------ forward
    def styles(self, latent):
        """(B, n_latent, S) -> all style vectors (B, R) in one contraction."""
        return ops.style_affine(latent, self.wcat, self.bcat, self.row_lat)

    def forward(self, latent, noises, save=False, cond_hook=None, cond_layers=None, return_features=False, features_in=None,
                feature_scale=1.0, range_mode='exact', post_hook=None):
        """latent (B,n_latent,S); noises list[num_layers] of (B|1,1,r,r).
        cond_hook(k, raw, latent_i, noise, noise_w) -> cond tensor replacing the raw up-conv output
        (the algebra of OOD_faceGAN_e4e_arch.py:239-242 + model.py:292: layer = cond + w*noise).
        features_in[i] (or None), feature_scale: `insert_feature` of the Feature-Style variant (model.py:541-546,557,572):
        the input of the styled conv that reads latent i becomes (1-fs)*x + fs*features_in[i].
        post_hook(k, out) -> tensor replacing the ACTIVATED output of the up-conv reading latent cond_layers[k]
        (feature_modulation for cond_type 'SFT' / 'ADD' / 'FUSE', model.py:558-566).
        range_mode (split-f16 only, ops.FwdRange): 'exact' measures max|x*s| of every conv input before converting it;
        'carry' (the W+ loop) uses the scales of the previous forward with the fused producers and verifies them."""
        if self.padded and (cond_hook is not None or features_in is not None or post_hook is not None):
            raise NotImplementedError('conditioning hooks / feature injection on a generator whose channel counts are zero-padded to multiples of 16')
        if save and (features_in is not None or post_hook is not None):
            raise NotImplementedError('backward through an injected feature / feature modulation is not part of the path')
        B = latent.shape[0]
        # forwards are counted: the activations saved for backward() live in pooled scratch buffers (ops.sform_scratch: keyed by shape and
        # stream) and their S-form scales alias the range state — ANY later forward of this engine on the same stream may overwrite them
        self._fwd_serial = getattr(self, '_fwd_serial', 0) + 1
        s_all = self.styles(latent)
        d_all = torch.empty(B, self.DR, device=self.device, dtype=torch.float32)
        if self.batched_tail:       # the demodulation factors of all styled convs in one launch
            from ._lib import DemodFwdJob, lib, check
            import ctypes
            styled = [L for L in self.layers if L.kind != 'rgb']
            arr = (DemodFwdJob * len(styled))(*[DemodFwdJob(ctypes.c_void_p(s_all.data_ptr() + 4 * L.row), ctypes.c_void_p(L.wsq.data_ptr()),
                                                            ctypes.c_void_p(d_all.data_ptr() + 4 * L.drow), self.R, self.DR, B, L.cin, L.cout,
                                                            float(L.scale)) for L in styled])
            check(lib().oodgan_demod_fwd_batch(arr, len(styled), ops._stream()), 'demod_fwd_batch')
        else:
            for L in self.layers:
                if L.kind != 'rgb':
                    self._demod(L, s_all, d_all)
        rng, carry = None, False
        s_use, d_use = s_all, d_all
        if self.sform:
            rng = self._range(B)
            # cond_hook (the SAMM hook of the OOD forward) does not stand in the way: a hooked layer's output is converted by to_s(),
            # which takes the carried scale and records the maximum like any fused producer
            carry = range_mode == 'carry' and rng.valid and self.carry_range and features_in is None and post_hook is None
            if carry:
                rng.plan(s_all, d_all)
            s_use, d_use = rng.s_sc, rng.d_sc

        def to_s(L, t):
            # S-form input of styled conv L from the fp32 tensor t (x its style x the layer's range scale)
            if not carry:
                rng.measure(L.sidx, t, _Cols(s_all, L.row, L.cin))
                rng.plan(s_all, d_all, L.row, L.cin, L.drow, L.cout)
            return ops.to_sform(t, _Cols(s_use, L.row, L.cin), out=ops.sform_scratch(B, L.cin, t.shape[2], t.shape[3], self.device),
                                vmax=rng.vm[L.sidx] if carry else None)

        acts = {}
        # ConstantInput broadcast over the batch (model.py:296-305): read-only, one copy per batch size — not a torch copy kernel per forward
        # (a launch plan, oodgan_plan_*, replays this library's launches only)
        cache = self.__dict__.setdefault('_const_b', {})
        ck = (B, ops._stream_handle())      # per stream: the copy is enqueued on the stream that first needs it
        x = cache.get(ck)
        if x is None:
            x = cache[ck] = self.const_input.expand(B, -1, -1, -1).contiguous()
        acts['input'] = x
        skip, out = None, x
        pending = None
        rgb_partial = None
        i = 1
        for L in self.layers:
            s = _Cols(s_all, L.row, L.cin)
            if L.kind == 'rgb' and rgb_partial is not None:
                skip = ops.rgb_finish(rgb_partial, L.bias, skip, self.k_up if skip is not None else None)
                rgb_partial = None
                continue
            if L.kind == 'rgb':
                Lu = self.rgb_next_up.get(L.name) if (carry and self.fused_rgb) else None
                if (Lu is not None and pending is None and out.shape[2] * out.shape[3] > 4096 and L.cin % 16 == 0
                        and out.shape[3] % 4 == 0):
                    # one pass over the feature map: RGB contribution AND the next up-conv's S-form input (x its style)
                    pending = ops.sform_scratch(B, L.cin, out.shape[2], out.shape[3], self.device)
                    skip = ops.torgb(out, L.w_rgb, s, L.bias, skip, self.k_up if skip is not None else None, ys=pending,
                                     ys_scale=_Cols(s_use, Lu.row, Lu.cin), vmax=rng.vm[Lu.sidx])
                else:
                    skip = ops.torgb(out, L.w_rgb, s, L.bias, skip, self.k_up if skip is not None else None)
                continue
            d = _Cols(d_use, L.drow, L.cout)
            nz = noises[L.noise_idx]
            if features_in is not None and L.lat >= 1 and L.lat < len(features_in) and features_in[L.lat] is not None:
                from . import samm as _samm
                f = features_in[L.lat].to(out.dtype).contiguous()
                C = out.shape[1]
                ones = torch.ones(B, C, device=self.device)
                zeros = torch.zeros(B, C, device=self.device)
                t = _samm.affine_apply(f, ones * float(feature_scale), zeros)
                out = _samm.affine_apply(out, ones * float(1.0 - feature_scale), zeros, res=t)
                pending = None              # an S-form written for the un-mixed tensor is stale
            if L.kind == 'conv':
                if self.sform:
                    # S-form hand-off: style folded in while splitting, the conv then streams its tiles by LDS-DMA
                    xf_in = isinstance(out, ops.FForm)      # the up-conv tail left its activation in F-form: converted inside the conv
                    if xf_in:
                        xs = out
                    elif pending is not None:     # written by the up-conv tail that produced `out`
                        xs, pending = pending, None
                    else:
                        xs = to_s(L, out)
                    Lr = self.conv_next_rgb.get(L.name) if self.fused_rgb else None
                    if Lr is not None and 16 < L.cin <= 32 and 16 < L.cout <= 32:
                        # 32-channel 1024² layer (strip kernel): the ToRGB colour sums come out of the same epilogue.
                        # Inside the W+ loop the activation of the LAST styled conv is read back by nothing but its own activation
                        # backward: once that runs as the fused producer (a carried scale exists) it is kept in F-form — both
                        # sides then move 16 bytes per lane on contiguous runs
                        ff = (save and carry and self.fused_bwd and L is self.layers_styled_last and L.cout == 32
                              and self.bwd_state.get(L.name) is not None and not return_features)
                        out, rgb_partial = ops.conv3x3(xs, L.wpk, L.cout, CONV_S1, out_scale=d, bias=L.bias, noise=nz,
                                                       noise_weight=L.noise_w, act=ACT_LRELU, rgb=(Lr.w_rgb, _Cols(s_all, Lr.row, Lr.cin)),
                                                       y_fform=ff or xf_in, in_scale=_Cols(s_use, L.row, L.cin) if xf_in else None)
                    else:
                        Lu = self.rgb_next_up.get(Lr.name) if (Lr is not None and carry) else None
                        Hc, Wc = xs.shape[2], xs.shape[3]
                        if (Lu is not None and self.fuse_conv_rgb and not xf_in and L.cout % 16 == 0 and Hc * Wc > (4095 if self.fuse_conv_rgb_64 else 4096) and Wc % 4 == 0
                                and ops.s1_ys_supported(B, L.cin, L.cout, Hc, Wc)):
                            # round 4: the 8-wave conv hands BOTH consumers of its output their input from its registers — the ToRGB colour
                            # sums (one partial per 64-channel block, finished by rgb_finish) and the S-form of out x style x range scale of
                            # the next up-conv, with the maximum for its range control: the torgb_fwd_sform pass (a read of the feature
                            # map at 128² ... 512²) disappears from the step
                            pending = ops.sform_scratch(B, L.cout, Hc, Wc, self.device, tag=4)
                            # both readers of the activation are served from the epilogue: without a backward pass to save it for, the fp32
                            # tensor is not written at all (the epilogue's stores are what these launches wait for, LABNOTES.md §13.9)
                            # ... and inside the W+ loop the saved activation exists ONCE: the only reader of this layer's output in the
                            # backward pass is the fused epilogue of the stride-2 conv above (style-gradient dot + this layer's
                            # activation backward), which decodes it from the same S-form (oodgan_conv_args.dotx_sform)
                            sform_only = (save and self.save_sform_only and self.fused_bwd and self.fuse_act_bwd
                                          and (self.sform_only_max_res is None or Hc <= self.sform_only_max_res)
                                          and self.bwd_state.get(L.name) is not None and self.bwd_state.get(Lu.name) is not None
                                          and ops.s2_fuse_supported(B, Lu.cout, Lu.cin, 2 * Hc + 1, 2 * Wc + 1))
                            keep = (save and not sform_only) or return_features or post_hook is not None
                            out, rgb_partial = ops.conv3x3(xs, L.wpk, L.cout, CONV_S1, out_scale=d, bias=L.bias, noise=nz, noise_weight=L.noise_w,
                                                           act=ACT_LRELU, rgb=(Lr.w_rgb, _Cols(s_all, Lr.row, Lr.cin)), ys=pending,
                                                           ys_scale=_Cols(s_use, Lu.row, Lu.cin), vmax=rng.vm[Lu.sidx], want_y=keep)
                            if not keep:
                                out = ops.SFormSaved(pending, _Cols(s_use, Lu.row, Lu.cin))
                        else:
                            out = ops.conv3x3(xs, L.wpk, L.cout, CONV_S1, out_scale=d, bias=L.bias, noise=nz,
                                              noise_weight=L.noise_w, act=ACT_LRELU)
                    del xs
                else:
                    out = ops.conv3x3(out, L.wpk, L.cout, CONV_S1, in_scale=s, out_scale=d, bias=L.bias, noise=nz,
                                      noise_weight=L.noise_w, act=ACT_LRELU)
            else:
                Hi = out.shape[2]
                Ln = self.next_conv.get(L.name)
                # last level inside the W+ loop: the activation stays in F-form and the conv converts it itself — no S-form copy of
                # the largest tensor of the step is written or read
                ff_tail = (self.sform and carry and self.fused_fwd and Ln is not None and save and self.fuse_x and self.fused_bwd and self.fused_rgb
                           and Ln is self.layers_styled_last and not return_features
                           and self.bwd_state.get(Ln.name) is not None and self.bwd_state.get(L.name) is not None
                           and ops.xf_supported(B, Ln.cin, Ln.cout, 2 * Hi, 2 * Hi))
                one_pass = ff_tail and self.fuse_up and L.wpk_vb is not None and ops.upconv_vblur_supported(B, L.cin, L.cout, Hi, Hi)
                # the plain forward (model(x), exact ranges) takes the same two kernels: the up-conv kernel records max|y * style| of the
                # F-form activation it writes, the range scale of the last conv is set from it on the device — no (2H+1)² intermediate,
                # no blur pass, no measurement pass and no S-form copy of the 1024² tensor
                plain_tail = (self.sform and not save and not one_pass and self.plain_one_pass and self.fuse_up and self.fuse_x
                              and self.fused_rgb and Ln is not None and Ln is self.layers_styled_last and L.wpk_vb is not None
                              and Ln.name in self.conv_next_rgb and 16 < Ln.cin <= 32 and 16 < Ln.cout <= 32
                              and not (cond_layers is not None and L.lat in cond_layers)
                              and not (features_in is not None and Ln.lat < len(features_in) and features_in[Ln.lat] is not None)
                              and ops.upconv_vblur_supported(B, L.cin, L.cout, Hi, Hi) and ops.xf_supported(B, Ln.cin, Ln.cout, 2 * Hi, 2 * Hi))
                z = None
                if self.sform:
                    if pending is not None:
                        xs, pending = pending, None
                    else:
                        xs = to_s(L, out)
                    if one_pass:
                        # ... and the (2H+1)² transposed-conv result is not written either: conv + blur + noise + bias + activation in ONE kernel
                        out = ops.upconv_vblur_fform(xs, L.wpk_vb, out_scale=d, bias=L.bias, noise=nz, noise_weight=L.noise_w, act=True,
                                                     ys_scale=_Cols(s_use, Ln.row, Ln.cin), vmax=rng.vm[Ln.sidx])
                    elif plain_tail:
                        out = ops.upconv_vblur_fform(xs, L.wpk_vb, out_scale=d, bias=L.bias, noise=nz, noise_weight=L.noise_w, act=True,
                                                     ys_scale=_Cols(s_use if carry else s_all, Ln.row, Ln.cin), vmax=rng.vm[Ln.sidx])
                        if not carry:       # exact ranges: the scale of the last conv from the maximum just recorded
                            rng.update_exact(Ln.sidx)
                            rng.plan(s_all, d_all, Ln.row, Ln.cin, Ln.drow, Ln.cout)
                        one_pass = True
                    else:
                        z = ops.conv3x3(xs, L.wpk, L.cout, CONV_T2, out_scale=d)
                    del xs
                else:
                    z = ops.conv3x3(out, L.wpk, L.cout, CONV_T2, in_scale=s, out_scale=d)
                H2 = 2 * Hi + 1
                lat_idx = L.lat
                if one_pass:
                    pass
                elif cond_hook is not None and cond_layers is not None and lat_idx in cond_layers:
                    raw = ops.blur_bias_act(z, self.k4x4, (1, 1), act=False, in_hw=(H2, H2), in_pitch=z.shape[3])
                    cond = cond_hook(cond_layers.index(lat_idx), raw, latent[:, lat_idx], nz, L.noise_w)
                    out = ops.bias_noise_act(cond, L.bias, nz, L.noise_w)
                elif carry and self.fused_fwd and L.name in self.next_conv:
                    # blur + noise + bias + activation, and the following conv's S-form input (x its style), in one pass
                    if ff_tail:
                        out = ops.blur_act_fform(z, self.k4x4, Hi, Hi, L.bias, nz, L.noise_w, act=True,
                                                 ys_scale=_Cols(s_use, Ln.row, Ln.cin), vmax=rng.vm[Ln.sidx], rank_one=self.k4x4_rank1)
                    else:
                        pending = ops.sform_scratch(B, L.cout, 2 * Hi, 2 * Hi, self.device, tag=2)
                        out = ops.blur_act_sform(z, self.k4x4, Hi, Hi, L.bias, nz, L.noise_w, act=True, ys=pending,
                                                 ys_scale=_Cols(s_use, Ln.row, Ln.cin), vmax=rng.vm[Ln.sidx], rank_one=self.k4x4_rank1)
                else:
                    out = ops.blur_bias_act(z, self.k4x4, (1, 1), L.bias, nz, L.noise_w, act=True, in_hw=(H2, H2),
                                            in_pitch=z.shape[3])
                del z
                if post_hook is not None and cond_layers is not None and lat_idx in cond_layers:
                    out = post_hook(cond_layers.index(lat_idx), out)
                    pending = None
            acts[L.name] = out
        if rng is not None:
            if carry:
                rng.finish()            # verify the carried scales, publish the next ones
            else:
                rng.valid = True
        if save:
            self.saved = dict(acts=acts, s_all=s_all, d_all=d_all, noises=noises, B=B, serial=self._fwd_serial)
        if return_features:
            feat = out.to_nchw() if isinstance(out, ops.FForm) else out
            return skip, (feat[:, :self.real_channels[self.size]].contiguous() if self.padded else feat)
        return skip

    def _demod(self, L, s_all, d_all):
        from ._lib import lib, check
        import ctypes
        B = s_all.shape[0]
        check(lib().oodgan_demod_fwd(ctypes.c_void_p(s_all.data_ptr() + 4 * L.row), self.R, ctypes.c_void_p(L.wsq.data_ptr()),
                                     ctypes.c_void_p(d_all.data_ptr() + 4 * L.drow), self.DR, B, L.cin, L.cout, L.scale,
                                     ops._stream()), 'demod_fwd')

    def _hi_records(self, B, L, Hd, Rg):
        """precision 'f16s-g2': the gradient of an up-conv layer goes to its stride-2 input-gradient conv as 32-byte hi-only records when that
        conv is the two-instruction 8-wave kernel (which never reads a lo half) and the producer is the strip walk (oodgan_act_bwd_blurT_sform_phases_hi):
        half the bytes written by the one and read by the other."""
        return bool(self.grad_hi_only and self.hi_records and Rg is None and ops.blurT_hi_supported(Hd, Hd)
                    and ops.s2_fuse_supported(B, L.cout, L.cin, 2 * Hd + 1, 2 * Hd + 1))

    # ------------------------------------------------------------------ backward (w.r.t. latents only)
    def backward(self, gimg, grad_scale=1.0, carry_scale=False):
        """gimg (B,3,size,size), already multiplied by ``grad_scale`` -> dL/dlatent (B,n_latent,S).
        Needs forward(..., save=True).  Every step is linear in the gradient, so a power-of-two
        grad_scale is undone exactly at the end.
        ``carry_scale`` (set by the W+ loop): from the second call on, the activation gradients are produced directly
        in the matrix kernels' input layout with the range scale measured on the previous call (fused producers);
        the caller must start a new sequence with reset_bwd_state() and check bwd_scale_violated() at its end."""
        from ._lib import lib, check
        import ctypes
        sv = self.saved
        if sv is None:
            raise RuntimeError('backward() without forward(save=True)')
        if sv.get('serial') != getattr(self, '_fwd_serial', None):
            raise RuntimeError('backward(): another forward of this engine ran after forward(save=True) — the saved activations (pooled S-form / F-form '
                               'buffers, scales of the range state) may have been overwritten; call backward() right after the forward it belongs to')
        acts, s_all, d_all, noises, B = sv['acts'], sv['s_all'], sv['d_all'], sv['noises'], sv['B']
        gs_all = ops.zeros(B, self.R, device=self.device)
        # gradient of the skip chain: gskip[res] for every ToRGB level
        gskip = {self.size: gimg.contiguous()}
        r = self.size
        while r > 4:
            gskip[r // 2] = ops.upfirdn2d(gskip[r], self.k_up_flip, up=1, down=2, pad=(1, 1))
            r //= 2
        g_feat = None
        prev_rgb = None
        fused_in = None         # ops.ActBwdFusion produced by the stride-2 conv of the layer above
        fused_pre = None        # ops.DotActGrad: the stride-1 conv above already applied this up-conv layer's act'
        jobs = ops.BwdJobs() if (carry_scale and self.fused_bwd and self.sform and self.batched_tail) else None
        for L in reversed(self.layers):
            if L.kind == 'rgb':          # fused into the backward of the styled conv that feeds it
                prev_rgb = L
                continue
            out, x_in, nz = acts[L.name], acts[L.src], noises[L.noise_idx]
            fused_ok = carry_scale and self.fused_bwd and self.bwd_state.get(L.name) is not None
            # last level: out and x_in in F-form -> the activation backward runs inside the input-gradient conv (ops.ActBwdX)
            xf_bwd = (fused_ok and g_feat is None and L.kind == 'conv' and isinstance(out, ops.FForm) and isinstance(x_in, ops.FForm)
                      and prev_rgb is not None and self.fuse_x)
            if isinstance(out, ops.FForm) and not (fused_ok and (g_feat is None or fused_pre is not None)):
                out = out.to_nchw()         # the two-pass path reads NCHW
            if isinstance(out, ops.SFormSaved) and fused_in is None:
                out = out.to_nchw()         # saved only as its consumer's S-form, but this layer's activation backward did not run fused
            if isinstance(x_in, ops.FForm) and not xf_bwd:
                x_in = x_in.to_nchw()
            s = _Cols(s_all, L.row, L.cin)
            d = _Cols(d_all, L.drow, L.cout)
            Hd = x_in.shape[2]
            carry = carry_scale and self.fused_bwd and self.sform and (L.kind == 'conv' or Hd >= 4)
            st = self.bwd_state.get(L.name) if carry else None
            Rg, prev_rgb = prev_rgb, None
            rgb_kw = {} if Rg is None else dict(g_rgb=gskip[L.res], w_rgb=Rg.w_rgb, s_rgb=_Cols(s_all, Rg.row, Rg.cin))
            if st is not None and fused_in is not None:
                # this layer's activation backward already ran in the epilogue of the stride-2 conv above (its S-form
                # gradient, partial sums and maxima come from there; g_feat never went to HBM)
                gin, rsum, tsum, part_m = fused_in.dst, fused_in.r, None, fused_in.part_m
                fused_in = None
                mul2, g_pre = st, None
            elif st is not None and xf_bwd:
                gin, rsum, tsum, part_m = None, None, None, None        # produced by the conv below (ops.ActBwdX)
                mul2, g_pre = st, None
            elif st is not None:
                # fused producer: g_pre goes straight into the next matrix kernel's input layout, scaled with the
                # range scale measured on the previous step (verified below, after the conv has consumed it)
                t_into = None if Rg is None else _Cols(gs_all, Rg.row, Rg.cin)
                if L.kind == 'conv':
                    gin = ops.sform_scratch(B, L.cout, out.shape[2], out.shape[3], self.device)
                    rsum, tsum, part_m = ops.act_bwd_producer(out, g_feat, nz, L.noise_w, L.bias, d, st, gin, t_into=t_into, jobs=jobs, **rgb_kw)
                elif fused_pre is not None:
                    # g_feat already is g_pre: blur^T + phase split of one tensor; r = noise / bias term + s * dot of that conv
                    gin = ops.sform_phases_scratch(B, L.cout, Hd, Hd, self.device)
                    rsum, tsum, part_m = ops.act_bwd_producer(None, g_feat, nz, L.noise_w, L.bias, d, st, gin, blur_kernel=self.k4x4_flip,
                                                              jobs=jobs, dot_of=fused_pre, hi_only=self._hi_records(B, L, Hd, Rg))
                    fused_pre = None
                else:
                    gin = ops.sform_phases_scratch(B, L.cout, Hd, Hd, self.device)
                    rsum, tsum, part_m = ops.act_bwd_producer(out, g_feat, nz, L.noise_w, L.bias, d, st, gin,
                                                              blur_kernel=self.k4x4_flip, t_into=t_into, jobs=jobs,
                                                              hi_only=self._hi_records(B, L, Hd, Rg), **rgb_kw)
                mul2, g_pre = st, None
            else:
                if Rg is not None:
                    g_pre, rsum, tsum, mul2 = ops.act_bwd_fused(out, g_feat, nz, L.noise_w, L.bias, gskip[L.res], Rg.w_rgb,
                                                                _Cols(s_all, Rg.row, Rg.cin), want_scale=True, dscale=d)
                else:
                    g_pre, rsum, tsum, mul2 = ops.act_bwd_fused(out, g_feat, nz, L.noise_w, L.bias, want_scale=True, dscale=d)
                if carry:
                    self.bwd_state[L.name] = mul2            # exact this step; carried to the next one
                    if self.bwd_flag is None:
                        self.bwd_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
            if Rg is not None and tsum is not None:
                gs_all[:, Rg.row:Rg.row + Rg.cin] = tsum
            # demodulation gradient
            deferred = st is not None and jobs is not None
            xa = None
            if st is not None and xf_bwd:
                xa = ops.ActBwdX(nz, L.noise_w, L.bias, d, st, gskip[L.res], Rg.w_rgb, _Cols(s_all, Rg.row, Rg.cin),
                                 t_into=_Cols(gs_all, Rg.row, Rg.cin))
                pre = None
                Lp = self.by_name.get(L.src)
                if (Lp is not None and Lp.kind == 'up' and self.fuse_act_bwd and self.bwd_state.get(L.src) is not None
                        and ops.s1_actgrad_supported(B, L.cout, L.cin, out.shape[2], out.shape[3])):
                    pre = ops.DotActGrad()
                dx, dot = ops.conv3x3(out, L.wpk_bwd, L.cin, CONV_S1, out_scale=s, dotx=x_in, in_mul2=mul2,
                                      dot_into=_Cols(gs_all, L.row, L.cin), jobs=jobs, dot_actgrad=pre, xf_act=xa)
                rsum, part_m = xa.r, xa.part_m
                fused_in, fused_pre = None, pre
            if deferred:        # rsum is filled by the batched reduction at the end of the pass; so is this job's input
                jobs.add_demod(_Cols(s_all, L.row, L.cin), L.wsq, _Cols(d_all, L.drow, L.cout), rsum, _Cols(gs_all, L.row, L.cin), B, L.cin,
                               L.cout, L.scale)
            else:
                check(lib().oodgan_demod_bwd(ctypes.c_void_p(s_all.data_ptr() + 4 * L.row), self.R, ctypes.c_void_p(L.wsq.data_ptr()),
                                             ctypes.c_void_p(d_all.data_ptr() + 4 * L.drow), self.DR, ctypes.c_void_p(rsum.data_ptr()),
                                             ctypes.c_void_p(gs_all.data_ptr() + 4 * L.row), self.R, B, L.cin, L.cout, L.scale,
                                             ops._stream()), 'demod_bwd')
            if xa is not None:
                if deferred:
                    jobs.add_check(part_m, st)
                else:
                    ops.absmax_scale_check(part_m, st, self.bwd_flag)
            elif st is not None:
                fz = None
                Lp = self.by_name.get(L.src)
                stp = self.bwd_state.get(L.src) if (L.kind == 'up' and Lp is not None and self.fuse_act_bwd) else None
                if stp is not None and ops.s2_fuse_supported(B, L.cout, L.cin, 2 * Hd + 1, 2 * Hd + 1):
                    # the stride-2 conv's result is the gradient w.r.t. the output of the conv layer below (x_in): its
                    # epilogue continues with that layer's activation backward (+ ToRGB branch) and writes the S-form
                    Rp = self.conv_next_rgb[Lp.name]
                    # f16s-g2: that S-form as 32-byte hi-only records when the conv that reads it is the two-instruction 8-wave stride-1 kernel
                    hi = bool(self.grad_hi_only and self.hi_records and ops.s1_xh_supported(B, Lp.cout, Lp.cin, Hd, Hd))
                    fz = ops.ActBwdFusion((ops.sform_hi_scratch if hi else ops.sform_scratch)(B, Lp.cout, Hd, Hd, self.device), noises[Lp.noise_idx],
                                          Lp.noise_w, Lp.bias, _Cols(d_all, Lp.drow, Lp.cout), stp, g_rgb=gskip[Lp.res], w_rgb=Rp.w_rgb,
                                          s_rgb=_Cols(s_all, Rp.row, Rp.cin), t_into=_Cols(gs_all, Rp.row, Rp.cin), hi_only=hi)
                pre = None
                if (L.kind == 'conv' and Lp is not None and Lp.kind == 'up' and self.fuse_act_bwd and self.bwd_state.get(L.src) is not None
                        and ops.s1_actgrad_supported(B, L.cout, L.cin, out.shape[2], out.shape[3])):
                    # x_in is the output of the up-sampling layer below: this conv's epilogue applies that layer's act',
                    # the blur^T producer then reads one tensor
                    pre = ops.DotActGrad()
                dx, dot = ops.conv3x3(gin, L.wpk_bwd, L.cin, CONV_S1 if L.kind == 'conv' else CONV_S2, out_scale=s, dotx=x_in,
                                      in_mul2=mul2, dot_into=_Cols(gs_all, L.row, L.cin), jobs=jobs, fuse=fz, want_y=fz is None,
                                      dot_actgrad=pre)
                fused_in, fused_pre = fz, pre
                del gin
                if deferred:
                    jobs.add_check(part_m, st)
                else:
                    ops.absmax_scale_check(part_m, st, self.bwd_flag)
            elif L.kind == 'conv':
                if self.sform:
                    gs_ = ops.to_sform(g_pre, d, mul2, out=ops.sform_scratch(B, L.cout, g_pre.shape[2], g_pre.shape[3], self.device))
                    dx, dot = ops.conv3x3(gs_, L.wpk_bwd, L.cin, CONV_S1, out_scale=s, dotx=x_in, in_mul2=mul2)
                    del gs_
                else:
                    dx, dot = ops.conv3x3(g_pre, L.wpk_bwd, L.cin, CONV_S1, in_scale=d, out_scale=s, dotx=x_in, in_mul2=mul2)
            else:
                if self.sform and Hd >= 4:
                    # blur^T, demodulation scale, range scale, phase split and f16 split in one pass, then the
                    # stride-2 conv as stride-1 taps on the four parity images
                    gp = ops.blurT_to_sform_phases(g_pre, self.k4x4_flip, d, mul2,
                                                   out=ops.sform_phases_scratch(B, L.cout, Hd, Hd, self.device))
                    dx, dot = ops.conv3x3(gp, L.wpk_bwd, L.cin, CONV_S2, out_scale=s, dotx=x_in, in_mul2=mul2)
                else:
                    H2 = 2 * Hd + 1
                    P2 = (H2 + 3) // 4 * 4
                    g2 = ops.upfirdn2d(g_pre, self.k4x4_flip, pad=(2, 2), out_pitch=P2)
                    dx, dot = ops.conv3x3(g2, L.wpk_bwd, L.cin, CONV_S2, in_scale=d, out_scale=s, dotx=x_in, in_hw=(H2, H2),
                                          in_pitch=P2, in_mul2=mul2)
                    del g2
            if dot is not None:
                gs_all[:, L.row:L.row + L.cin] += dot
            g_feat = dx
            del g_pre
        if jobs is not None:
            jobs.run(self.bwd_flag)         # four launches: partial sums, demodulation gradient, dot products, scale checks
        self.last_gs = gs_all
        return ops.style_affine_backward(gs_all, self.wcat, self.lat_start, self.n_latent, grad_div=grad_scale)


_SIDE = {}


def _side_streams(device, n):
    """The same n side streams on every call: the caching allocator's pools and the S-form scratch buffers are keyed by
    stream, so fresh streams per inversion would mean fresh (synchronising) device allocations inside every inversion."""
    pool = _SIDE.setdefault(str(device), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


# Concurrent sub-batches (streams > 1): the 8-wave stride-1 kernel from 64 work items instead of 128.  A sub-batch of four images brings 64 items
# to the 32² layers; on the 8-wave kernel they occupy 64 CUs for a short time and leave the rest of the chip to the other stream's launch, the
# 4-wave ring kernel it replaces spreads 256 small workgroups over every CU.  Measured (bench.py, 2 streams, three boxes): 6.97-6.99 -> 7.10-7.16
# img/s; on ONE stream (batch 8) the lower threshold moves the 16² layers and costs 1.3 % — hence only here (LABNOTES.md §14.6).
MULTI_STREAM_S1_BIG_MIN_ITEMS = int(os.environ.get('OODGAN_MULTI_STREAM_S1_BIG_MIN_ITEMS', '64'))
_S1_BIG_DEFAULT = 128

_FLAG = {}
_PLAN_POOLS = {}


def _plan_pool(device, batch):
    """The private allocator pool a launch plan is recorded under, ONE per (device, stream, batch size) for the life of the process: the buffers a
    recorded step touched keep their addresses while the plan lives, and the next inversion's recording reuses the same blocks (a fresh pool per
    inversion hands ~5 GB back to the driver and takes it again: 35 ms per inversion of 8, measured).  Per batch size: a step of ONE image recorded
    into blocks a batch of 8 left behind ran 2.6x slower (724 against 274 ms per inversion, tools/plan_pool_probe.py) — its tensors are then carved out
    of a few multi-GB segments instead of sized segments."""
    if os.environ.get('OODGAN_PLAN_POOL_CACHE', '1') == '0':
        return torch.cuda.MemPool()
    key = (str(device), ops._stream_handle(), int(batch))
    pool = _PLAN_POOLS.get(key)
    if pool is None:
        pool = _PLAN_POOLS[key] = torch.cuda.MemPool()
    return pool


def _flag_stream(device):
    """The stream the range flags are read on (one per device): a 4-byte device-to-host copy behind an event of the compute stream."""
    st = _FLAG.get(str(device))
    if st is None:
        st = _FLAG[str(device)] = torch.cuda.Stream(device=device)
    return st


class _WRun:
    """One (sub-)batch of the W+ loop on one HIP stream, advanced one step per ``advance()``.

    Range guard (round 6): the carried forward / backward range scales raise sticky device flags when a tensor's range
    moved by more than the format's head-room within one step (range control).  Until round 5 the flags were read after the last
    step and the WHOLE inversion repeated with exact per-step scales.  Now, every ``check_every`` steps, (w, m, v, t) is snapshotted and the
    two flags are copied to pinned host memory on a side stream behind an event of the compute stream — no synchronisation of the compute
    stream; ``check_lag`` steps later (the copy has long finished: the host runs ahead of the GPU) the host looks at them.  Clear: the
    snapshot becomes the last clean state.  Set: the run goes back to the last clean state, repeats that window (<= check_every steps)
    with exact scales, and continues in carry mode from freshly measured scales."""

    def __init__(self, inv, eng, target, w0, noises, steps, stream, dev_counter, keep_traj):
        self.inv, self.eng, self.stream, self.steps = inv, eng, stream, steps
        self.target, self.noises = target, noises
        self.w = w0.detach().clone().contiguous()
        self.m, self.v = torch.zeros_like(self.w), torch.zeros_like(self.w)
        self.gmul = ops.loss_scale_for(target.numel() // target.shape[0])
        self.t = 0
        self.dev_t = torch.zeros(1, dtype=torch.int32, device=self.w.device) if dev_counter else None
        self.lbuf = torch.empty(steps, self.w.shape[0], device=self.w.device, dtype=torch.float32)
        # optional LPIPS(alex) term (oodgan/lpips.py): the target's normalised taps once, a second loss table
        self.lp = inv.lpips if (inv.lpips is not None and inv.lpips_weight != 0.0) else None
        self.lp_target = self.lp.target_taps(target) if self.lp is not None else None
        self.lp_table = torch.zeros(steps, self.w.shape[0], device=self.w.device, dtype=torch.float32) if self.lp is not None else None
        self.traj = [None] * steps if keep_traj else None
        self.mode0 = (eng.fused_bwd, eng.carry_range)            # what the caller asked for (tests run the exact loop on purpose)
        self.guard = (eng.fused_bwd or eng.carry_range) and inv.check_every > 0
        self.clean = (0, self.w.clone(), None, None) if self.guard else None     # m = v = 0 at t = 0
        self.pending, self.exact_until = None, 0
        self.steps_run = 0                                       # forward/backward pairs enqueued, repeated windows included
        self.rollbacks = 0
        self.host = inv._pinned(len(inv._runs)) if self.guard else None
        inv._runs.append(self)
        # launch plan (round 6): the first step measures exact scales (its own launch sequence); the second — the first
        # steady-state step — runs from Python as well: it CREATES the persistent scratch buffers only the carried-scale path uses (pooled
        # S-forms with a zero border, hi-only record buffers, max slots).  Created inside the recording they would be carved out of blocks
        # that temporaries of the same step held earlier, and on every replay those temporaries' kernels would write over them — borders
        # included (seen as NaN from the second replay on).  The third step is RECORDED while it runs (oodgan_plan_*, under a private
        # allocator pool so that every buffer it touched keeps its address); every later step is one oodgan_plan_run call instead of ~100
        # ctypes calls and the Python between them
        self.use_plan = bool(inv.use_plan) and dev_counter and not keep_traj and eng.sform
        self.plan, self.pool, self.eager_left, self.plan_steps, self.plan_size = None, None, 2, 0, 0

    # ---- one W+ step on the current stream
    def _eager_step(self):
        eng, inv = self.eng, self.inv
        img = eng.forward(self.w, self.noises, save=True, range_mode='carry')
        if self.dev_t is not None:
            # loss row and Adam's step index from the device counter: the recorded step is the same launch list for every t
            _, gimg = ops.mse_loss_grad(img, self.target, self.gmul, table=self.lbuf, row_dev=self.dev_t)
            if self.lp is not None:     # gimg += gmul * lambda * d(sum_b lpips_b)/d(img); values to row t of the second table
                self.lp.loss_and_grad(img, gimg, inv.lpips_weight * self.gmul, table=self.lp_table, row_dev=self.dev_t, target_taps=self.lp_target)
            g = eng.backward(gimg, self.gmul, carry_scale=True)
            ops.adam_step_dev(self.w, g, self.m, self.v, self.dev_t, inv.lr, inv.betas, inv.eps)
        else:
            _, gimg = ops.mse_loss_grad(img, self.target, self.gmul, loss_out=self.lbuf[self.t])
            if self.lp is not None:
                self.lp_table[self.t].copy_(self.lp.loss_and_grad(img, gimg, inv.lpips_weight * self.gmul, target_taps=self.lp_target))
            g = eng.backward(gimg, self.gmul, carry_scale=True)
            ops.adam_step(self.w, g, self.m, self.v, self.t + 1, inv.lr, inv.betas, inv.eps)

    def _step(self):
        eng = self.eng
        if self.plan is not None:
            self.plan.run()
            self.plan_steps += 1
        elif (self.use_plan and self.eager_left == 0 and not self.exact_until and eng.fused_bwd and eng.carry_range
              and self.steps - self.t >= 3):
            plan, pool = ops.LaunchPlan(), _plan_pool(self.w.device, self.w.shape[0])
            with torch.cuda.use_mem_pool(pool):
                with plan.recording():
                    self._eager_step()
            self.plan, self.pool, self.plan_size = plan, pool, plan.size
        else:
            self._eager_step()
            self.eager_left = max(0, self.eager_left - 1)
        self.t += 1
        self.steps_run += 1
        if self.traj is not None:
            self.traj[self.t - 1] = self.w.clone()
        if self.inv.on_step is not None:
            self.inv.on_step(self)

    def _post_check(self):
        eng = self.eng
        snap = (self.t, self.w.clone(), self.m.clone(), self.v.clone())
        ev = torch.cuda.Event()
        ev.record()
        side, done = _flag_stream(self.w.device), torch.cuda.Event()
        self.host.zero_()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            if eng.bwd_flag is not None:
                self.host[0:1].copy_(eng.bwd_flag, non_blocking=True)
            if eng.fwd_range is not None:
                self.host[1:2].copy_(eng.fwd_range.flag, non_blocking=True)
            done.record(side)
        self.pending = (snap, done)

    def _resolve(self):
        """Look at the flags of the pending check; True if the run was rolled back."""
        snap, done = self.pending
        self.pending = None
        done.synchronize()
        if int(self.host[0]) == 0 and int(self.host[1]) == 0:
            self.clean = snap
            return False
        eng = self.eng
        t0, w, m, v = self.clean
        self.w.copy_(w)
        if m is None:
            self.m.zero_()
            self.v.zero_()
        else:
            self.m.copy_(m)
            self.v.copy_(v)
        self.t = t0
        if self.dev_t is not None:
            self.dev_t.fill_(t0)
        self.rollbacks += 1
        self.plan = None                                         # the carried state it points to is rebuilt: recorded again after the window
        eng.fused_bwd = eng.carry_range = False                  # exact per-step scales, both directions, for the flagged window
        eng.reset_bwd_state()
        eng.reset_fwd_state()
        self.exact_until = snap[0]
        return True

    def _leave_exact(self):
        eng = self.eng
        eng.fused_bwd, eng.carry_range = self.mode0
        eng.reset_bwd_state()                                    # the next step measures its scales exactly, like the first of a run
        eng.reset_fwd_state()
        self.exact_until = 0
        self.eager_left = 2
        self.clean = (self.t, self.w.clone(), self.m.clone(), self.v.clone())

    def _advance(self):
        if self.t < self.steps:
            self._step()
            if self.exact_until:
                if self.t == self.exact_until:
                    self._leave_exact()
            elif self.guard and (self.t % self.inv.check_every == 0 or self.t == self.steps):
                if self.pending is not None and self._resolve():     # (only when the last step follows a check within the lag)
                    return
                self._post_check()
        if self.pending is not None and (self.t >= self.pending[0][0] + self.inv.check_lag or self.t >= self.steps):
            self._resolve()

    def advance(self):
        """Enqueue the next step (or resolve a check / roll back); False once the run is complete and verified."""
        if self.done():
            return False
        if self.stream is None:
            self._advance()
        else:
            with torch.cuda.stream(self.stream):
                self._advance()
        return not self.done()

    def done(self):
        return self.t >= self.steps and self.pending is None

    def close(self):
        self.eng.fused_bwd, self.eng.carry_range = self.mode0
        self.eng.saved = None
        self.plan = None            # before the pool it points into
        self.pool = None


class WPlusInverter:
    """Build-defined W+ optimisation loop: ``steps`` x {G(w) with fixed noise,
    per-image MSE, backward to w, Adam(lr, betas, eps)} — anchors: reference Generator.forward with
    ``noise=<list>`` (model.py:483-585), torch.optim.Adam as built by get_optimizer
    (src/models/OOD_faceGAN_model.py:398-400), basicsr MSELoss (losses.py:58-83).

    ``check_every`` / ``check_lag``: the range guard of ``_WRun`` (0 = flags read only by the caller).  ``last_stats`` after a call:
    {'steps_run': forward/backward pairs enqueued per (sub-)batch, 'rollbacks': windows repeated with exact scales}."""

    def __init__(self, engine, lr=0.01, betas=(0.9, 0.999), eps=1e-8, check_every=10, check_lag=2, use_plan=None, lpips=None, lpips_weight=0.0):
        self.engine, self.lr, self.betas, self.eps = engine, lr, betas, eps
        # loss = per-image MSE + lpips_weight * LPIPS(alex) (north_star: "W+ Adam steps against LPIPS/L2"); ``lpips``: an oodgan.lpips.LPIPSAlex
        # (min_max = the generator's output range).  Off by default: the `lpips` weights are third-party and absent here (parity unpinned).
        # ``invert`` then returns the TOTAL loss per step and image; ``last_terms`` = {'mse', 'lpips'} tables.
        self.lpips, self.lpips_weight = lpips, float(lpips_weight)
        self.last_terms = None
        # launch plans (oodgan_plan_*): on unless OODGAN_USE_PLAN=0; the single-stream loop uses the device step counter with them
        self.use_plan = (os.environ.get('OODGAN_USE_PLAN', '1') != '0') if use_plan is None else bool(use_plan)
        self.check_every, self.check_lag = int(check_every), max(0, int(check_lag))
        self._runs, self._pins = [], []
        self.last_stats = self.last_plan = None
        self.on_step = None         # optional callable(run) after every enqueued step (progress reporting; tests inject faults with it)

    def _pinned(self, i):
        while len(self._pins) <= i:
            self._pins.append(torch.zeros(2, dtype=torch.int32).pin_memory())
        return self._pins[i]

    def invert(self, target, w0, noises, steps=100, return_trajectory=False, streams=1, use_graph=False):
        """``streams`` > 1 splits the batch into that many independent sub-batches, each advanced on its own HIP
        stream (images are independent): the HBM-bound layout/activation kernels of one sub-batch
        then share the GPU with the matrix kernels of the other instead of running back to back."""
        B = w0.shape[0]
        streams = max(1, min(int(streams), B))
        if B == 0:          # an empty shard (oodgan/parallel.py hands a rank with no images an empty batch): nothing to launch
            steps = 0
        if steps <= 0:
            w = w0.detach().clone().contiguous()
            empty = torch.empty(0, B, device=w0.device, dtype=torch.float32)
            return (w, empty, []) if return_trajectory else (w, empty)
        if return_trajectory:
            streams, use_graph = 1, False
        if use_graph:
            return self._invert_graph(target, w0, noises, steps, streams)
        if streams == 1:
            return self._invert_runs(target, w0, noises, steps, 1, return_trajectory)
        from . import _lib
        # the multi-stream threshold of the 8-wave stride-1 kernel, unless the caller set the tunable itself; the value that was read is
        # the one restored
        old = _lib.lib().oodgan_get_tunable(b's1_big_min_items')
        override = old == _S1_BIG_DEFAULT and MULTI_STREAM_S1_BIG_MIN_ITEMS != _S1_BIG_DEFAULT
        if override:
            _lib.set_tunable('s1_big_min_items', MULTI_STREAM_S1_BIG_MIN_ITEMS)
        try:
            return self._invert_runs(target, w0, noises, steps, streams, False)
        finally:
            if override:
                _lib.set_tunable('s1_big_min_items', old)

    def _invert_runs(self, target, w0, noises, steps, streams, return_trajectory):
        B = w0.shape[0]
        self._runs = []
        if streams == 1:
            eng = self.engine
            eng.reset_bwd_state()
            eng.reset_fwd_state()
            runs = [_WRun(self, eng, target, w0, noises, steps, None, self.use_plan and not return_trajectory, return_trajectory)]
        else:
            cur = torch.cuda.current_stream()
            side = _side_streams(w0.device, streams)
            # the state resets enqueue zero-fills on the CALLER's stream: they must precede the wait_stream below, or nothing
            # orders them against the atomicOr / atomic max the side streams do on the same flags (engines[0] keeps its
            # FwdRange and bwd_flag across calls)
            engines = [self.engine] + [self.engine.clone_shared() for _ in range(streams - 1)]
            for eng in engines:
                eng.reset_bwd_state()
                eng.reset_fwd_state()
            cuts = [(i * B) // streams for i in range(streams + 1)]
            parts = []
            for i in range(streams):
                sl = slice(cuts[i], cuts[i + 1])
                parts.append((target[sl].contiguous(), w0[sl].detach().contiguous(),
                              [n[sl].contiguous() if n.shape[0] == B else n for n in noises]))
            for st in side:
                st.wait_stream(cur)                 # AFTER the slices above were enqueued on the caller's stream
            runs = []
            for i, st in enumerate(side):
                with torch.cuda.stream(st):         # state tensors initialised on the stream that uses them
                    runs.append(_WRun(self, engines[i], parts[i][0], parts[i][1], parts[i][2], steps, st, True, False))
        try:
            busy = True
            while busy:
                busy = False
                for r in runs:
                    busy = r.advance() or busy
        finally:
            for r in runs:
                r.close()
        self.last_stats = {'steps_run': [r.steps_run for r in runs], 'rollbacks': [r.rollbacks for r in runs]}
        self.last_plan = {'steps': [r.plan_steps for r in runs], 'launches': [r.plan_size for r in runs]}
        if streams == 1:
            r = runs[0]
            losses = self._total(r.lbuf, r.lp_table)
            return (r.w, losses, r.traj) if return_trajectory else (r.w, losses)
        cur = torch.cuda.current_stream()
        for st in side:
            cur.wait_stream(st)
        w = torch.cat([r.w for r in runs], 0)
        losses = torch.cat([r.lbuf for r in runs], 1)
        lp = torch.cat([r.lp_table for r in runs], 1) if runs[0].lp_table is not None else None
        for r in runs:                      # tensors produced on side streams are consumed on the caller's stream
            r.w.record_stream(cur)
            r.lbuf.record_stream(cur)
            if r.lp_table is not None:
                r.lp_table.record_stream(cur)
        return w, self._total(losses, lp)

    def _total(self, mse, lp):
        self.last_terms = {'mse': mse, 'lpips': lp}
        return mse if lp is None else mse + self.lpips_weight * lp

    def _invert_graph(self, target, w0, noises, steps, streams):
        """hipGraph replay of one captured W+ step per stream (measured 3 % slower than eager launches on this host; kept as an
        option).  A replayed step cannot change its arithmetic mid-run: the range flags are read at the end and a flagged inversion is
        repeated eagerly (with the per-window guard)."""
        if self.lpips is not None and self.lpips_weight != 0.0:
            raise NotImplementedError('use_graph with the LPIPS term: use the launch plans (default) instead')
        B = w0.shape[0]
        cur = torch.cuda.current_stream()
        side = _side_streams(w0.device, streams)
        engines = [self.engine] + [self.engine.clone_shared() for _ in range(streams - 1)]
        for eng in engines:
            eng.reset_bwd_state()
            eng.reset_fwd_state()
        cuts = [(i * B) // streams for i in range(streams + 1)]
        parts = []
        for i, st in enumerate(side):
            sl = slice(cuts[i], cuts[i + 1])
            parts.append(dict(target=target[sl].contiguous(), w=w0[sl].detach().clone().contiguous(),
                              noises=[n[sl].contiguous() if n.shape[0] == B else n for n in noises]))
            parts[-1]['m'] = torch.zeros_like(parts[-1]['w'])
            parts[-1]['v'] = torch.zeros_like(parts[-1]['w'])
        for st in side:
            st.wait_stream(cur)
        gmul = ops.loss_scale_for(target.numel() // B)
        dev = w0.device

        def one_step(pr, eng):
            img = eng.forward(pr['w'], pr['noises'], save=True, range_mode='carry')
            loss, gimg = ops.mse_loss_grad(img, pr['target'], gmul)
            g = eng.backward(gimg, gmul, carry_scale=True)
            ops.adam_step_dev(pr['w'], g, pr['m'], pr['v'], pr['t'], self.lr, self.betas, self.eps)
            return loss

        for i, st in enumerate(side):
            pr = parts[i]
            with torch.cuda.stream(st):
                pr['t'] = torch.zeros(1, dtype=torch.int32, device=dev)
                pr['lbuf'] = torch.empty(steps, pr['w'].shape[0], device=dev, dtype=torch.float32)
        for i, st in enumerate(side):       # step 1 eagerly (also warms allocator pools / scratch buffers of every stream) ...
            with torch.cuda.stream(st):
                parts[i]['lbuf'][0].copy_(one_step(parts[i], engines[i]))
        graphs = [None] * streams
        if steps > 1:
            # ... then ONE W+ step is captured per stream into a hipGraph and replayed
            for i, st in enumerate(side):
                st.synchronize()
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=st):
                    parts[i]['lstat'] = one_step(parts[i], engines[i])
                graphs[i] = gph
        for t in range(2, steps + 1):
            for i, st in enumerate(side):
                with torch.cuda.stream(st):
                    graphs[i].replay()
                    parts[i]['lbuf'][t - 1].copy_(parts[i]['lstat'])
        for st in side:
            cur.wait_stream(st)
        for eng in engines:
            eng.saved = None
        w = torch.cat([pr['w'] for pr in parts], 0)
        losses = torch.cat([pr['lbuf'] for pr in parts], 1)
        for pr in parts:
            pr['w'].record_stream(cur)
            pr['lbuf'].record_stream(cur)
        self._graphs = graphs               # keep the graphs (and their private pools) alive until the next call
        self.last_stats = {'steps_run': [steps] * streams, 'rollbacks': [0] * streams}
        if any(eng.bwd_scale_violated() or eng.fwd_range_violated() for eng in engines):
            return self.invert(target, w0, noises, steps, False, streams, False)
        return w, losses
