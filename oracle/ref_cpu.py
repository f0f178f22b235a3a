"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional, plain-torch-CPU restatement of the reference algorithm for the hot path
(SURVEY.md §8 rows A1-A11).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker /
reported baseline.  The product path (``ood-gan-inversion_amd/``) never imports it.

Pinning: the reference holds no test or golden vector for this path, so the
oracle is pinned against outputs of the reference itself, produced in the build container by
``tests/golden/make_golden.py`` (which imports ``/root/reference`` with stub modules) and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those vectors.  Unpinned: LPIPS (third-party ``lpips`` package, no version
pinned anywhere in the reference, weights absent) — the W+ loop is pinned on L2 only.

Every function cites the reference lines it follows.  Parameters are passed as a flat dict
``P`` keyed with the reference's state-dict names (rosinality layout).

The convolutions issue the same ATen graph as the reference (materialised per-sample
weights + grouped ``conv2d`` / ``conv_transpose2d``; pad + ``conv2d`` upfirdn) so that timing
this module is a fair "reference CPU path" baseline.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = 2 ** 0.5


# --------------------------------------------------------------------------- L1 custom ops
def make_kernel(taps):
    """reference src/ops/StyleGAN/model.py:19-27."""
    k = torch.as_tensor(taps, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Zero-stuff by ``up``, pad (negative pad crops), true convolution with ``kernel``
    (i.e. correlation with the flipped kernel), keep every ``down``-th sample.
    reference src/ops/op/upfirdn2d.py:149-193 (native branch; same (pad0,pad1) on x and y).
    out = (in*up + pad0 + pad1 - k)//down + 1."""
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    p0, p1 = int(pad[0]), int(pad[1])
    v = x.reshape(B * C, 1, H, W)
    if up > 1:
        z = v.new_zeros(B * C, 1, H * up, W * up)
        z[:, :, ::up, ::up] = v
        v = z
    v = F.pad(v, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    c0, c1 = max(-p0, 0), max(-p1, 0)
    v = v[:, :, c0:v.shape[2] - c1, c0:v.shape[3] - c1]
    v = F.conv2d(v, torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(v.dtype))
    v = v[:, :, ::down, ::down]
    return v.reshape(B, C, v.shape[2], v.shape[3])


def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=SQRT2):
    """scale * leaky_relu(x + bias[channel]) — reference src/ops/op/fused_act.py:92-96."""
    if bias is not None:
        x = x + bias.reshape((1, -1) + (1,) * (x.ndim - 2))
    return scale * F.leaky_relu(x, negative_slope=negative_slope)


# --------------------------------------------------------------------------- A1 style affine / MLP
def equal_linear(x, weight, bias=None, lr_mul=1.0, activation=False):
    """reference model.py:129-158: W stored /lr_mul, scale = lr_mul/sqrt(in), bias*lr_mul."""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        out = F.linear(x, weight * scale)
        return fused_leaky_relu(out, bias * lr_mul)
    return F.linear(x, weight * scale, bias=None if bias is None else bias * lr_mul)


def mapping_network(P, z, n_mlp=8, lr_mlp=0.01, prefix=''):
    """PixelNorm + n_mlp x EqualLinear(lr_mul, fused_lrelu) — reference model.py:11-16,391-400."""
    h = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(1, n_mlp + 1):
        h = equal_linear(h, P[f'{prefix}style.{i}.weight'], P[f'{prefix}style.{i}.bias'], lr_mlp, True)
    return h


# --------------------------------------------------------------------------- A2/A3 modulated conv
def modulated_conv2d(x, w_lat, weight, mod_weight, mod_bias, demodulate=True, upsample=False,
                     blur_taps=(1, 3, 3, 1), blur_kernel=None):
    """reference model.py:233-274 (plain and upsample branches; downsample is off-path).

    x (B,Ci,H,W); w_lat (B,style_dim); weight (1,Co,Ci,k,k).  Literal 1e-8 at :240."""
    B, Ci, H, W = x.shape
    _, Co, _, k, _ = weight.shape
    s = equal_linear(w_lat, mod_weight, mod_bias).reshape(B, 1, Ci, 1, 1)
    wgt = (1.0 / math.sqrt(Ci * k * k)) * weight * s
    if demodulate:
        d = torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8)
        wgt = wgt * d.reshape(B, Co, 1, 1, 1)
    if upsample:
        wt = wgt.transpose(1, 2).reshape(B * Ci, Co, k, k)
        y = F.conv_transpose2d(x.reshape(1, B * Ci, H, W), wt, padding=0, stride=2, groups=B)
        y = y.reshape(B, Co, y.shape[2], y.shape[3])
        # Blur(pad=(pad0,pad1), upsample_factor=2): model.py:199-205,72-88
        # ``blur_kernel``: the registered buffer ``conv.blur.kernel`` (model.py:72-81) as the checkpoint holds it — a loaded state dict
        # overrides what the constructor's ``blur_kernel`` taps built; the pads follow the tap count either way
        kb = make_kernel(blur_taps) * 4.0 if blur_kernel is None else blur_kernel
        p = (kb.shape[0] - 2) - (k - 1)
        pad0, pad1 = (p + 1) // 2 + 2 - 1, p // 2 + 1
        return upfirdn2d(y, kb, pad=(pad0, pad1))
    y = F.conv2d(x.reshape(1, B * Ci, H, W), wgt.reshape(B * Co, Ci, k, k), padding=k // 2, groups=B)
    return y.reshape(B, Co, y.shape[2], y.shape[3])


def styled_conv(P, prefix, x, w_lat, noise, upsample=False, hook=None):
    """conv -> noise injection -> bias + lrelu*sqrt2.  reference model.py:343-350,283-292.

    ``hook(raw_conv_out, w_lat, noise, noise_weight)`` (if given) plays the role of the
    reference callback at model.py:288-290: it returns the tensor that replaces ``noise``."""
    y = modulated_conv2d(x, w_lat, P[f'{prefix}.conv.weight'], P[f'{prefix}.conv.modulation.weight'],
                         P[f'{prefix}.conv.modulation.bias'], True, upsample,
                         blur_kernel=P.get(f'{prefix}.conv.blur.kernel') if upsample else None)
    nw = P[f'{prefix}.noise.weight']
    if hook is not None:
        noise = hook(y, w_lat, noise, nw)
    y = y + nw * noise
    return fused_leaky_relu(y, P[f'{prefix}.activate.bias'])


def to_rgb(P, prefix, x, w_lat, skip=None, blur_taps=(1, 3, 3, 1)):
    """1x1 modulated conv without demod + bias (+ 2x upsampled skip). reference model.py:353-372,30-48."""
    y = modulated_conv2d(x, w_lat, P[f'{prefix}.conv.weight'], P[f'{prefix}.conv.modulation.weight'],
                         P[f'{prefix}.conv.modulation.bias'], demodulate=False)
    y = y + P[f'{prefix}.bias']
    if skip is not None:
        ku = P.get(f'{prefix}.upsample.kernel')          # Upsample's registered buffer (model.py:30-48), as loaded
        if ku is None:
            ku = make_kernel(blur_taps) * 4.0
        p = ku.shape[0] - 2
        y = y + upfirdn2d(skip, ku, up=2, pad=((p + 1) // 2 + 1, p // 2))
    return y


# --------------------------------------------------------------------------- Generator (A11 + loop)
def feature_modulation(gen_feats, conditions, mod_type='SFT'):
    """reference model.py:588-610 with clss=None (-> 1); FUSE does not mutate ``conditions`` here."""
    if mod_type == 'SFT':
        return gen_feats * (1 + conditions[0]) + conditions[1]
    if mod_type == 'ADD':
        return gen_feats + conditions[1]
    if mod_type == 'FUSE':
        return gen_feats + conditions[1] * torch.sigmoid(conditions[0])
    raise NotImplementedError(f'unknown mod_type {mod_type}')


def generator_forward(P, latent, noises, size, prefix='', cond_layers=None, hook=None,
                      return_features=False, post_hook=None):
    """``Generator.forward(latent, input_is_tensor=True, input_is_latent=True, noise=noises, ...)``.
    reference model.py:548-585.  ``latent`` (B, n_latent, style_dim); ``noises`` list of
    (B|1,1,r,r).  For layer index i in ``cond_layers`` the up-conv runs with
    ``hook(k, raw, w_lat, noise, nw)``, k = cond_layers.index(i) (model.py:558-569)."""
    B = latent.shape[0]
    log_size = int(math.log2(size))
    feats = []
    out = P[f'{prefix}input.input'].repeat(B, 1, 1, 1)
    out = styled_conv(P, f'{prefix}conv1', out, latent[:, 0], noises[0])
    feats.append(out)
    skip = to_rgb(P, f'{prefix}to_rgb1', out, latent[:, 1])
    i = 1
    for j in range(log_size - 2):
        h = None
        if cond_layers is not None and hook is not None and i in cond_layers:
            k = cond_layers.index(i)
            h = (lambda raw, wl, nz, nw, _k=k: hook(_k, raw, wl, nz, nw))
        out = styled_conv(P, f'{prefix}convs.{2 * j}', out, latent[:, i], noises[2 * j + 1], True, h)
        if cond_layers is not None and post_hook is not None and i in cond_layers:      # cond_type != 'NOISE' (model.py:561-564)
            out = post_hook(cond_layers.index(i), out)
        feats.append(out)
        out = styled_conv(P, f'{prefix}convs.{2 * j + 1}', out, latent[:, i + 1], noises[2 * j + 2])
        feats.append(out)
        skip = to_rgb(P, f'{prefix}to_rgbs.{j}', out, latent[:, i + 2], skip)
        i += 2
    if return_features:
        return skip, feats
    return skip


# --------------------------------------------------------------------------- A7 SAMM / SAIM
def instance_norm(x, weight=None, bias=None, eps=1e-5):
    """nn.InstanceNorm2d (biased variance, eps 1e-5, never running stats)."""
    return F.instance_norm(x, None, None, weight, bias, True, 0.1, eps)


def bottleneck_ir(P, prefix, x):
    """bottleneck_IR(in, depth, stride=1, bn='InstanceNorm', bias=False) —
    reference src/ops/e4e/encoders/helpers.py:426-448."""
    w1 = P[f'{prefix}.res_layer.1.weight']
    r = instance_norm(x, P[f'{prefix}.res_layer.0.weight'], P[f'{prefix}.res_layer.0.bias'])
    r = F.conv2d(r, w1, padding=1)
    r = F.prelu(r, P[f'{prefix}.res_layer.2.weight'])
    r = F.conv2d(r, P[f'{prefix}.res_layer.3.weight'], padding=1)
    r = instance_norm(r, P[f'{prefix}.res_layer.4.weight'], P[f'{prefix}.res_layer.4.bias'])
    if f'{prefix}.shortcut_layer.0.weight' in P:
        sc = F.conv2d(x, P[f'{prefix}.shortcut_layer.0.weight'])
        sc = instance_norm(sc, P[f'{prefix}.shortcut_layer.1.weight'], P[f'{prefix}.shortcut_layer.1.bias'])
    else:
        sc = x  # MaxPool2d(1, 1) is the identity
    return r + sc


def align_net(P, prefix, source, target, scale, diff_fAndg=True):
    """AlignNet.forward(source, target) — reference SAMM/helpers.py:96-109 (``diff_fAndg``: :98-101).
    Returns (B,3,H,W) = [tanh*scale, tanh*scale, sigmoid]."""
    s, t = instance_norm(source), instance_norm(target)
    a = torch.cat([s - t, t], dim=1) if diff_fAndg else torch.cat([s, t], dim=1)
    a = bottleneck_ir(P, f'{prefix}.body.0', a)
    a = bottleneck_ir(P, f'{prefix}.body.1', a)
    return torch.cat([torch.tanh(a[:, 0:1]) * scale, torch.tanh(a[:, 1:2]) * scale,
                      torch.sigmoid(a[:, 2:])], dim=1)


def new_prm(x, y):
    """reference SAMM/helpers.py:62-77 (g = x): y*up(x) + up(x)*(1-up(x)), bicubic align_corners=True."""
    if x.shape[-2:] != y.shape[-2:]:
        x = F.interpolate(x, size=y.shape[-2:], mode='bicubic', align_corners=True)
    return y * x + x * (1 - x)


def spm_add(aligned, align, scale):
    """SPM_Warp.add — reference SAMM/helpers.py:129-137."""
    dx = torch.clip(aligned[:, 0:1] + align[:, 0:1], -scale, scale)
    dy = torch.clip(aligned[:, 1:2] + align[:, 1:2], -scale, scale)
    al = torch.clip(new_prm(aligned[:, 2:], align[:, 2:]), 0.0, 1.0)
    return torch.cat([dx, dy, al], dim=1)


def spm_upsample_add(aligned, align):
    """SPM_Warp.upsample_add — reference SAMM/helpers.py:139-147."""
    al = torch.clip(new_prm(aligned[:, 2:], align[:, 2:]), 0.0, 1.0)
    return torch.cat([align[:, 0:1], align[:, 1:2], al], dim=1)


def warp_blend(target, field):
    """grid = identity(linspace -1..1, 'ij') + (dx,dy); grid_sample(bilinear, zeros,
    align_corners=False); lerp with alpha — reference SAMM/helpers.py:168-177."""
    B, _, H, W = target.shape
    ys = torch.linspace(-1, 1, H, device=target.device)
    xs = torch.linspace(-1, 1, W, device=target.device)
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    grid = torch.stack([gx.unsqueeze(0) + field[:, 0], gy.unsqueeze(0) + field[:, 1]], dim=-1)
    warped = F.grid_sample(target, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    alpha = field[:, 2:]
    return warped * alpha + target * (1 - alpha)


def spm_warp(P, prefix, source, target, aligned=None, scale=0.08, cycle_align=2, diff_fAndg=True):
    """SPM_Warp.forward(source=encoder feat, target=generator feat, aligned=coarser field).
    reference SAMM/helpers.py:149-179.  The blur is ``Blur(pad=(2,1))`` with the un-scaled 4x4."""
    blur_k = make_kernel((1, 3, 3, 1))
    cur = target
    acc = None
    for k in range(cycle_align):
        a = upfirdn2d(align_net(P, f'{prefix}.body', cur, source, scale, diff_fAndg), blur_k, pad=(2, 1))
        acc = a if acc is None else spm_add(acc, a, scale)
        if k == cycle_align - 1 and aligned is not None:
            acc = spm_upsample_add(aligned, acc)
        cur = warp_blend(target, acc)
    return cur, acc


def blending_mask(aligns, size):
    """reference OOD_faceGAN_e4e_arch.py:315-339 (note ``a_k*a + a*(1-a)`` at :333)."""
    alpha = None
    for key in sorted(k for k in aligns if k != size):
        a_k = F.interpolate(aligns[key][:, 2:], size=(size, size), mode='bilinear')
        alpha = a_k if alpha is None else a_k * alpha + alpha * (1 - alpha)
    return None if alpha is None else torch.clip(alpha, 0.0, 1.0)


def ood_forward(P, x, enc_lats, enc_feats, noises, size=1024, warp_scale=0.08, cycle_align=2,
                truncation=1.0, blend_with_gen=True):
    """Everything of ``ood_faceGAN_e4e.forward`` after the e4e encoder call.
    reference OOD_faceGAN_e4e_arch.py:263-313 with feats2condition_callback :224-242.

    enc_lats (B,18,512) and enc_feats[0..3] ((B,64,256²),(B,64,128²),(B,128,64²),(B,256,32²))
    are the encoder outputs (psp_encoders.py:178-214).  Returns (out, lats, aligns)."""
    lats = enc_lats + P['avg_latent'].reshape(1, 1, -1) + P['delta_latent']
    if truncation < 1.0:
        lats = P['avg_latent'].reshape(1, 1, -1) * (1.0 - truncation) + lats * truncation
    feats = [F.conv2d(enc_feats[i], P[f'feats_conv.{i}.weight'], P[f'feats_conv.{i}.bias']) for i in range(4)]
    aligns = {}

    def hook(k, raw, w_lat, noise, nw):
        ind = k + 1
        cond, field = spm_warp(P, f'modulation.{4 - ind}.alignment', feats[-ind], raw,
                               aligns.get(ind - 1), warp_scale, cycle_align)
        aligns[ind] = field
        return (cond - raw + noise * nw) / nw

    gen = generator_forward(P, lats, noises, size, 'generator.', [5, 7, 9, 11], hook)
    out = gen
    if blend_with_gen:
        alpha = blending_mask(aligns, size)
        aligns[size] = alpha.repeat(1, 3, 1, 1)
        out = alpha * x + gen * (1 - alpha)
    return out, lats, aligns


def extract_masks(aligns):
    """reference run_ood_faceGAN_inversion.py:74-87: alpha channel of every level, nearest
    upsample to 1024, concatenated along width."""
    parts = [F.interpolate(aligns[k][:, 2:], size=(1024, 1024)) for k in sorted(aligns)]
    return torch.cat(parts, dim=-1)


# --------------------------------------------------------------------------- A9 W+ optimisation loop
def wplus_loss(img, target):
    """Sum over the batch of per-image mean squared error (basicsr MSELoss 'mean' applied per
    image, BasicSR/basicsr/losses/losses.py:58-83) so that every image's trajectory is
    independent of the batch it is sharded into."""
    return ((img - target) ** 2).mean(dim=(1, 2, 3)).sum()


def wplus_invert(P, target, w0, noises, size, steps=100, lr=0.01, betas=(0.9, 0.999), eps=1e-8,
                 prefix='', return_trajectory=False):
    """Build-defined W+ loop: torch autograd through the restated generator
    with fixed noise, ``torch.optim.Adam`` as constructed by the reference's get_optimizer
    (src/models/OOD_faceGAN_model.py:398-400).  Returns (w, losses[steps])."""
    w = w0.detach().clone().requires_grad_(True)
    opt = torch.optim.Adam([w], lr=lr, betas=betas, eps=eps)
    losses, traj = [], []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        img = generator_forward(P, w, noises, size, prefix)
        per_img = ((img - target) ** 2).mean(dim=(1, 2, 3))
        per_img.sum().backward()
        losses.append(per_img.detach().clone())
        opt.step()
        if return_trajectory:
            traj.append(w.detach().clone())
    if return_trajectory:
        return w.detach(), torch.stack(losses), traj
    return w.detach(), torch.stack(losses)
