#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REAL REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  The reference is a
Python repo, so it is imported (never copied): four stub modules stand in for packages the
image lacks (torchvision, easydict, basicsr.utils / registry / arch_util — SURVEY.md
Appendix C).  Parameters come from ``oodgan.synth`` (numpy PCG64 keyed by state-dict key), so
the tests rebuild the same weights without the reference; only inputs that are small and all
expected outputs are stored.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, 'ood-gan-inversion_amd'))
sys.path.insert(0, '/root/reference')

from oodgan import synth  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    tv = _stub('torchvision')
    tv.ops = _stub('torchvision.ops', deform_conv2d=None)
    tv.transforms = _stub('torchvision.transforms')
    tv.models = _stub('torchvision.models', resnet34=None)
    tv.utils = _stub('torchvision.utils')

    class EasyDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    _stub('easydict', EasyDict=EasyDict)

    class Registry:
        def __init__(self, n):
            self._d = {}

        def register(self, obj=None):
            if obj is None:
                return lambda o: self._d.setdefault(o.__name__, o)
            self._d[obj.__name__] = obj
            return obj

        def get(self, k):
            return self._d[k]

    _stub('basicsr')
    _stub('basicsr.utils', scandir=lambda *a, **k: iter(()))
    _stub('basicsr.utils.registry',
          **{f'{n}_REGISTRY': Registry(n) for n in ('ARCH', 'LOSS', 'METRIC', 'MODEL', 'DATASET')})
    _stub('basicsr.archs')
    _stub('basicsr.archs.arch_util', trunc_normal_=torch.nn.init.trunc_normal_)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, keys={len(out)}')


def gold_ops():
    """Per-op vectors: upfirdn2d, fused_leaky_relu, EqualLinear, ModulatedConv2d x3,
    StyledConv, ToRGB."""
    from src.ops.op import upfirdn2d, fused_leaky_relu
    from src.ops.StyleGAN.model import (EqualLinear, ModulatedConv2d, StyledConv, ToRGB, make_kernel)
    g = {}
    k4 = make_kernel([1, 3, 3, 1])
    x = synth.normal('ops.x', (2, 3, 9, 11), 11)
    g['ufd_x'] = x
    # the three parameterisations on the forward path + the two adjoints
    for tag, kern, up, down, pad in [
        ('blur11', k4 * 4, 1, 1, (1, 1)), ('up2', k4 * 4, 2, 1, (2, 1)), ('blur21', k4, 1, 1, (2, 1)),
        ('down2', k4 * 4, 1, 2, (1, 2)), ('blur22', k4 * 4, 1, 1, (2, 2)), ('crop', k4, 1, 1, (-1, 3)),
    ]:
        g[f'ufd_{tag}'] = upfirdn2d(x, kern, up=up, down=down, pad=pad)
    b = synth.normal('ops.b', (3,), 11)
    g['flr_b'] = b
    g['flr_y'] = fused_leaky_relu(x, b)
    g['flr_y2'] = fused_leaky_relu(x, b, negative_slope=0.1, scale=1.5)
    xl = synth.normal('ops.xl', (4, 32), 11)
    g['lin_x'] = xl
    for act in (None, 'fused_lrelu'):
        lin = EqualLinear(32, 24, bias_init=0.0, lr_mul=0.01, activation=act)
        lin.weight.data = synth.normal('ops.lin.w', (24, 32), 11, 100.0)
        lin.bias.data = synth.normal('ops.lin.b', (24,), 11)
        g['lin_w'], g['lin_b'] = lin.weight.data, lin.bias.data
        g['lin_y_act' if act else 'lin_y'] = lin(xl)
    # modulated convs, small channel counts
    B, Ci, Co, H, S = 2, 16, 8, 12, 64
    xm = synth.normal('ops.mc.x', (B, Ci, H, H), 12)
    wl = synth.normal('ops.mc.wlat', (B, S), 12)
    g['mc_x'], g['mc_wlat'] = xm, wl
    for tag, k, demod, ups in [('plain', 3, True, False), ('up', 3, True, True), ('rgb', 1, False, False)]:
        cout = 3 if tag == 'rgb' else Co
        mc = ModulatedConv2d(Ci, cout, k, S, demodulate=demod, upsample=ups)
        mc.weight.data = synth.normal(f'ops.mc.{tag}.w', (1, cout, Ci, k, k), 12)
        mc.modulation.weight.data = synth.normal(f'ops.mc.{tag}.mw', (Ci, S), 12)
        mc.modulation.bias.data = synth.normal(f'ops.mc.{tag}.mb', (Ci,), 12, 0.1, 1.0)
        g[f'mc_{tag}_w'], g[f'mc_{tag}_mw'], g[f'mc_{tag}_mb'] = mc.weight.data, mc.modulation.weight.data, mc.modulation.bias.data
        g[f'mc_{tag}_y'] = mc(xm, wl)
    # StyledConv (noise + bias + act) plain and upsample, ToRGB with skip
    for tag, ups in [('sc', False), ('scup', True)]:
        sc = StyledConv(Ci, Co, 3, S, upsample=ups)
        sd = {}
        synth._styled_conv(sd, 'q', Ci, Co, S, 13, ups, 0.1)
        sc.load_state_dict({k_[2:]: v for k_, v in sd.items()})
        r = 2 * H if ups else H
        nz = synth.normal(f'ops.{tag}.noise', (B, 1, r, r), 13)
        g[f'{tag}_noise'] = nz
        g[f'{tag}_y'] = sc(xm, wl, noise=nz)
    rgb = ToRGB(Ci, S)
    sd = {}
    synth._to_rgb(sd, 'q', Ci, S, 13, True)
    rgb.load_state_dict({k_[2:]: v for k_, v in sd.items()})
    skip = synth.normal('ops.rgb.skip', (B, 3, H // 2, H // 2), 13)
    g['rgb_skip'] = skip
    g['rgb_y'] = rgb(xm, wl, skip)
    g['rgb_y_noskip'] = rgb(xm, wl, None)
    save('ops.npz', **g)


def gold_generator(size=32):
    from src.ops.StyleGAN.model import Generator
    torch.manual_seed(0)
    G = Generator(size, 512, 8).eval()
    sd = synth.generator_state(size, seed=5)
    missing = G.load_state_dict(sd, strict=True)
    print('generator load:', missing)
    B = 2
    lat = synth.make_latents(size, B, seed=6)
    noises = synth.make_noises(size, B, seed=7)
    g = {}
    with torch.no_grad():
        img, feat = G(lat, input_is_tensor=True, input_is_latent=True, noise=noises, return_features=True)
        g['image'] = img
        g['last_feature_sub'] = feat[:, ::16]
        # stored noise buffers + mapping network path ([z] -> style MLP -> broadcast latent)
        z = synth.normal('gen.z', (B, 512), 8)
        img2, lat2 = G([z], randomize_noise=False, return_latents=True)
        g['z'] = z
        g['image_from_z'] = img2
        g['latent_from_z'] = lat2
        # truncation path
        mean_lat = synth.normal('gen.mean_lat', (1, 512), 8, 0.3)
        img3, _ = G([z], randomize_noise=False, truncation=0.7, truncation_latent=mean_lat)
        g['mean_lat'] = mean_lat
        g['image_trunc'] = img3
    save(f'generator_s{size}.npz', **g)
    return G, sd, lat, noises


def gold_generator_narrow(size=64, narrow=0.5):
    """``StyleGAN2Generator(out_size, narrow=0.5)`` of the REFERENCE (src/ops/StyleGAN/stylegan2_arch.py:399-605; channel counts x narrow, :422,435-443):
    forward from W+ latents with explicit noise.  Weights: the rosinality-layout recipe state at the narrowed channel counts, remapped to the BasicSR
    key layout by the test's own map (checked here against the reference module's key set)."""
    from src.ops.StyleGAN.stylegan2_arch import StyleGAN2Generator
    from oodgan.modules import StyleGAN2Generator as Mine
    G = StyleGAN2Generator(size, narrow=narrow).eval()
    ros = synth.generator_state(size, seed=5, narrow=narrow)
    mine = Mine(size, narrow=narrow)
    bsd = {mine._ros_to_basicsr(k): v for k, v in ros.items() if not k.endswith('.kernel')}
    missing = G.load_state_dict(bsd, strict=True)
    print('narrow generator load:', missing)
    B = 2
    lat = synth.make_latents(size, B, seed=6)
    noises = synth.make_noises(size, B, seed=7)
    with torch.no_grad():
        img, _ = G(lat, input_is_latent=True, noise=noises)          # with input_is_latent the reference takes the (B, n_latent, S) tensor itself
    st = max(size // 128, 1)        # 64²: the whole image; larger sizes: a sub-sample and moments
    save(f'generator_narrow_s{size}.npz', image=img[:, :, ::st, ::st], image_mean=img.double().mean(dim=(2, 3)), image_std=img.double().std(dim=(2, 3)),
         narrow=np.float64(narrow), channels=np.asarray([G.channels[str(2 ** i)] for i in range(2, int(np.log2(size)) + 1)]))


def gold_wplus(size=32, steps=5):
    """W+ Adam trajectory through the REFERENCE Generator autograd (SURVEY §8 A9 anchors)."""
    from src.ops.StyleGAN.model import Generator
    G = Generator(size, 512, 8).eval()
    G.load_state_dict(synth.generator_state(size, seed=5), strict=True)
    for p in G.parameters():
        p.requires_grad_(False)
    B = 2
    target = synth.make_images(size, B, seed=9)
    noises = synth.make_noises(size, B, seed=7)
    w = synth.make_latents(size, B, seed=6).clone().requires_grad_(True)
    opt = torch.optim.Adam([w], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    losses, traj, grads = [], [], []
    for _ in range(steps):
        opt.zero_grad()
        img, _ = G(w, input_is_tensor=True, input_is_latent=True, noise=noises)
        per = ((img - target) ** 2).mean(dim=(1, 2, 3))
        per.sum().backward()
        losses.append(per.detach().clone())
        grads.append(w.grad.detach().clone())
        opt.step()
        traj.append(w.detach().clone())
    save(f'wplus_s{size}.npz', losses=torch.stack(losses), traj=torch.stack(traj), grads=torch.stack(grads))


WPLUS_1024_IMAGE = 10     # global image index of the bench recipe (bench.py: weights seed 0, target 1000+g, noise 2000+g, latents 3000+g)


def gold_wplus_1024():
    """ONE W+ step at the benchmarked geometry (BASELINE configs[2]: 1024², the bench recipe's weights and the inputs of
    its global image 10) through the REFERENCE Generator's autograd, evaluated in float64 (`G.double()`: same reference
    code, no fp32 LeakyReLU-kink ambiguity) and in the reference's own float32.  Image 10 is the recipe image whose
    low-resolution pre-activations keep the widest margin from the kink (scratch scan: >= 3e-6 in the 4²..16² layers).
    Anchors: model.py:483-585, BasicSR/basicsr/losses/losses.py:58-83."""
    from src.ops.StyleGAN.model import Generator
    size, gi = 1024, WPLUS_1024_IMAGE
    sd = synth.generator_state(size, seed=0)
    target = synth.make_images(size, 1, seed=1000 + gi)
    noises = synth.make_noises(size, 1, seed=2000 + gi)
    lat = synth.make_latents(size, 1, seed=3000 + gi, std=0.3)
    g = {}
    for tag, dt in (('f64', torch.float64), ('f32', torch.float32)):
        G = Generator(size, 512, 8).eval()
        G.load_state_dict(sd, strict=True)
        G = G.to(dt)
        for p in G.parameters():
            p.requires_grad_(False)
        w = lat.to(dt).clone().requires_grad_(True)
        img, _ = G(w, input_is_tensor=True, input_is_latent=True, noise=[n.to(dt) for n in noises])
        per = ((img - target.to(dt)) ** 2).mean(dim=(1, 2, 3))
        per.sum().backward()
        g[f'loss_{tag}'] = per.detach()
        g[f'grad_{tag}'] = w.grad.detach()
        if tag == 'f64':
            im = img.detach()
            g['image_sub'] = im[:, :, ::16, ::16].float()
            g['image_crop'] = im[:, :, 480:544, 480:544].float()
            g['image_mean'] = im.mean(dim=(2, 3))
            g['image_std'] = im.std(dim=(2, 3))
            g['image_absmax'] = im.abs().amax()
        del G, img, per, w
    rel = (g['grad_f32'].double() - g['grad_f64']).abs().max() / g['grad_f64'].abs().max()
    print(f'wplus_1024: loss {g["loss_f64"].item():.6f}; reference fp32 vs fp64 gradient: rel {rel.item():.2e}')
    g['image_index'] = np.int64(gi)
    save('wplus_1024.npz', **g)


BLUR_TAPS = (1, 4, 2, 1)        # an ASYMMETRIC four-tap filter: a flipped or transposed kernel anywhere in the chain shows
UP_TAPS = (2, 1, 3, 1)          # ... and another one for ToRGB's Upsample where the fixture says so


def gold_wplus_blur(size, gidx, steps, up_taps=(1, 3, 3, 1)):
    """``Generator(size, 512, 8, blur_kernel=[1,4,2,1])`` of the reference (model.py:376-384: the taps of every Blur / Upsample): one forward
    and ``steps`` Adam steps through the reference's autograd in float64 (and the first step in its own float32) on the bench recipe's inputs.
    The filter is asymmetric on purpose — with [1,3,3,1] a flipped, transposed or mirrored kernel in a fused producer / its adjoint is invisible.
    ``up_taps``: the kernel of ToRGB's Upsample.  As CONSTRUCTED it stays [1,3,3,1] whatever ``blur_kernel`` says (model.py:455 builds ToRGB
    without it) — the 64² fixture; it is a registered buffer, so a checkpoint can hold another one (strict load) — the 256² / 1024² fixtures."""
    from src.ops.StyleGAN.model import Generator
    sd = synth.generator_state(size, seed=0, blur_kernel=BLUR_TAPS, upsample_kernel=up_taps)
    cat = lambda parts: torch.cat(parts, 0)
    target = cat([synth.make_images(size, 1, seed=1000 + g) for g in gidx])
    per_n = [synth.make_noises(size, 1, seed=2000 + g) for g in gidx]
    noises = [cat([n[i] for n in per_n]) for i in range(len(per_n[0]))]
    lat = cat([synth.make_latents(size, 1, seed=3000 + g, std=0.3) for g in gidx])
    g = {'taps': np.asarray(BLUR_TAPS, dtype=np.int64), 'up_taps': np.asarray(up_taps, dtype=np.int64), 'image_indices': np.asarray(gidx, dtype=np.int64)}
    for tag, dt in (('f64', torch.float64), ('f32', torch.float32)):
        G = Generator(size, 512, 8, blur_kernel=list(BLUR_TAPS)).eval()
        if tuple(up_taps) == (1, 3, 3, 1):      # the constructor alone already built these buffers
            assert all(torch.equal(G.state_dict()[k], sd[k]) for k in sd if k.endswith('.kernel'))
        G.load_state_dict(sd, strict=True)
        G = G.to(dt)
        for p in G.parameters():
            p.requires_grad_(False)
        w = lat.to(dt).clone().requires_grad_(True)
        opt = torch.optim.Adam([w], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
        losses, traj = [], []
        for t in range(steps if tag == 'f64' else 1):
            opt.zero_grad()
            img, _ = G(w, input_is_tensor=True, input_is_latent=True, noise=[n.to(dt) for n in noises])
            per = ((img - target.to(dt)) ** 2).mean(dim=(1, 2, 3))
            per.sum().backward()
            losses.append(per.detach().clone())
            if t == 0:
                g[f'grad_{tag}'] = w.grad.detach().clone()
                if tag == 'f64':
                    im = img.detach()
                    st = max(size // 64, 1)
                    g['image_sub'] = im[:, :, ::st, ::st].float()
                    g['image_mean'] = im.mean(dim=(2, 3))
                    g['image_std'] = im.std(dim=(2, 3))
                    g['image_absmax'] = im.abs().amax()
            opt.step()
            traj.append(w.detach().clone())
        if tag == 'f64':
            g['losses'] = torch.stack(losses)
            g['traj'] = torch.stack(traj)
        del G, img, per, w
    rel = (g['grad_f32'].double() - g['grad_f64']).abs().max() / g['grad_f64'].abs().max()
    print(f'wplus_blur_{size}: losses {g["losses"][:, 0].tolist()}; reference fp32 vs fp64 gradient: rel {rel.item():.2e}')
    save(f'wplus_blur_{size}.npz', **g)


def gold_generator_resample(size=64):
    """``StyleGAN2Generator(out_size, resample_kernel=(1,4,2,1))`` of the reference (stylegan2_arch.py:420,455-494: the taps reach every smoothing
    layer AND ToRGB's up-sampling; its kernels are plain attributes, not buffers): forward from W+ latents with explicit noise."""
    from src.ops.StyleGAN.stylegan2_arch import StyleGAN2Generator
    from oodgan.modules import StyleGAN2Generator as Mine
    G = StyleGAN2Generator(size, resample_kernel=BLUR_TAPS).eval()
    ros = synth.generator_state(size, seed=5)
    mine = Mine(size)
    G.load_state_dict({mine._ros_to_basicsr(k): v for k, v in ros.items() if not k.endswith('.kernel')}, strict=True)
    lat = synth.make_latents(size, 2, seed=6)
    noises = synth.make_noises(size, 2, seed=7)
    with torch.no_grad():
        img, _ = G(lat, input_is_latent=True, noise=noises)
    save(f'generator_resample_s{size}.npz', image=img, taps=np.asarray(BLUR_TAPS, dtype=np.int64))


WPLUS_LONG = {
    # name -> (size, global image indices of the bench recipe, steps, dtype tag, stored latent checkpoints)
    'wplus_long_256': (256, (0, 1), 100, 'f32', (1, 5, 10, 20, 30, 60, 100)),
    'wplus_long_256_f64': (256, (0, 1), 100, 'f64', (1, 5, 10, 20, 30, 60, 100)),
    'wplus_long_1024': (1024, (WPLUS_1024_IMAGE,), 100, 'f32', (1, 5, 10, 20, 30, 60, 100)),
    'wplus_long_1024_f64': (1024, (WPLUS_1024_IMAGE,), 40, 'f64', (1, 5, 10, 20, 30, 40)),
    # image 0 of the bench batch (bench.py compares the per-image final loss of its timed inversion with this curve's end)
    'wplus_long_1024_img0': (1024, (0,), 100, 'f32', (10, 30, 60, 100)),
    # ... from the latents bench.py's inversion STARTS from: encoder latents + avg_latent + delta_latent of synth.ood_state(1024, seed 0)
    # (OOD_faceGAN_e4e_arch.py:264-267) — bench.py checks image 0 of its timed batch against this curve ('loss_check' in its JSON line)
    'wplus_long_1024_bench0': (1024, (0,), 100, 'f32', (10, 30, 60, 100)),
}


def gold_wplus_long(name):
    """The METRIC'S OWN HORIZON (BASELINE: "100 W+ steps"): the reference ``Generator``'s autograd + ``torch.optim.Adam``
    (anchors model.py:483-585, src/models/OOD_faceGAN_model.py:398-400, BasicSR losses.py:58-83) for 100 steps on the
    bench recipe (weights seed 0, target 1000+g, noise 2000+g, latents 3000+g with std 0.3) — at 256² for images (0, 1),
    at 1024² for image 10 — in the reference's own float32 and, as the yardstick for how far two correct implementations
    drift apart, in float64 (same reference code through ``G.double()``).  Stored: the per-image loss of every step, the
    latents at a few steps, dL/dW+ of the first and the last step, statistics of the image G(w_final).  Partial results are
    written after every checkpoint step so that an interrupted run still leaves a usable (shorter) fixture."""
    from src.ops.StyleGAN.model import Generator
    size, gidx, steps, tag, cps = WPLUS_LONG[name]
    dt = torch.float64 if tag == 'f64' else torch.float32
    B = len(gidx)
    G = Generator(size, 512, 8).eval()
    G.load_state_dict(synth.generator_state(size, seed=0), strict=True)
    G = G.to(dt)
    for p in G.parameters():
        p.requires_grad_(False)
    cat = lambda parts: torch.cat(parts, 0)
    target = cat([synth.make_images(size, 1, seed=1000 + g) for g in gidx]).to(dt)
    per_n = [synth.make_noises(size, 1, seed=2000 + g) for g in gidx]
    noises = [cat([n[i] for n in per_n]).to(dt) for i in range(len(per_n[0]))]
    w = cat([synth.make_latents(size, 1, seed=3000 + g, std=0.3) for g in gidx]).to(dt)
    if name.endswith('bench0'):
        ood = synth.ood_state(size, seed=0)
        w = w + ood['avg_latent'].reshape(1, 1, -1).to(dt) + ood['delta_latent'].to(dt)
    w = w.clone().requires_grad_(True)
    opt = torch.optim.Adam([w], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    g = {'image_indices': np.asarray(gidx, dtype=np.int64), 'steps': np.int64(steps)}
    losses = []
    import time
    t0 = time.time()
    for t in range(1, steps + 1):
        opt.zero_grad()
        img, _ = G(w, input_is_tensor=True, input_is_latent=True, noise=noises)
        per = ((img - target) ** 2).mean(dim=(1, 2, 3))
        per.sum().backward()
        losses.append(per.detach().double().clone())
        if t in (1, steps):
            g[f'grad_step{t}'] = w.grad.detach().double().clone()
        opt.step()
        if t in cps:
            g[f'w_step{t}'] = w.detach().double().clone()
            g['losses'] = torch.stack(losses)
            print(f'{name}: step {t}/{steps} loss {losses[-1].tolist()} ({time.time() - t0:.0f}s)', flush=True)
            save(name + '.npz', **g)
        del img, per
    with torch.no_grad():
        img, _ = G(w, input_is_tensor=True, input_is_latent=True, noise=noises)
        per = ((img - target) ** 2).mean(dim=(1, 2, 3))
    g['losses'] = torch.stack(losses)
    g['final_loss'] = per.double()
    g['final_image_sub'] = img[:, :, ::max(size // 64, 1), ::max(size // 64, 1)].float()
    g['final_image_mean'] = img.double().mean(dim=(2, 3))
    g['final_image_std'] = img.double().std(dim=(2, 3))
    save(name + '.npz', **g)


sys.path.insert(0, os.path.dirname(HERE))
from make_golden_params import GEN_B4  # noqa: E402


def gold_generator_1024_b4():
    """BASELINE configs[1] (C2): the StyleGAN2 1024² generator forward at batch 4 in the reference's own fp32 —
    ``Generator([z], noise=<list>)``: mapping MLP (model.py:391-400) -> broadcast latent -> model.py:483-585 — on the bench
    recipe's weights (seed 0).  Stored per image: a ::16 sub-sample, one 64x64 crop, per-channel mean / std, absmax, and
    the latent after the mapping network (inputs are rebuilt from the seeds by the test)."""
    from src.ops.StyleGAN.model import Generator
    size, B = GEN_B4['size'], GEN_B4['batch']
    sd = synth.generator_state(size, seed=0)
    G = Generator(size, 512, 8).eval()
    G.load_state_dict(sd, strict=True)
    z = synth.normal('gen_b4.z', (B, 512), GEN_B4['z_seed'])
    noises = synth.make_noises(size, B, seed=GEN_B4['noise_seed'])
    g = {}
    with torch.no_grad():
        img, lat = G([z], noise=noises, return_latents=True)
        img64, _ = G.double()([z.double()], noise=[n.double() for n in noises])
    g['latent'] = lat[:, 0]
    g['image_sub'] = img[:, :, ::16, ::16]
    g['image_crop'] = img[:, :, 448:512, 512:576]
    g['image_mean'] = img.double().mean(dim=(2, 3))
    g['image_std'] = img.double().std(dim=(2, 3))
    g['image_absmax'] = img.abs().amax()
    g['ref_f32_vs_f64'] = (img.double() - img64).abs().max()
    print(f'generator_1024_b4: absmax {g["image_absmax"].item():.3f}; reference fp32 vs its own float64: {g["ref_f32_vs_f64"].item():.2e}')
    save('generator_1024_b4.npz', **g)


def gold_wplus_256(steps=5):
    """5-step W+ Adam trajectory at 256² (B=2) through the reference Generator autograd (fp32, as shipped)."""
    from src.ops.StyleGAN.model import Generator
    size, B = 256, 2
    G = Generator(size, 512, 8).eval()
    G.load_state_dict(synth.generator_state(size, seed=0), strict=True)
    for p in G.parameters():
        p.requires_grad_(False)
    target = synth.make_images(size, B, seed=71)
    noises = synth.make_noises(size, B, seed=72)
    w = synth.make_latents(size, B, seed=73, std=0.3).clone().requires_grad_(True)
    opt = torch.optim.Adam([w], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    losses, traj, grads = [], [], []
    for _ in range(steps):
        opt.zero_grad()
        img, _ = G(w, input_is_tensor=True, input_is_latent=True, noise=noises)
        per = ((img - target) ** 2).mean(dim=(1, 2, 3))
        per.sum().backward()
        losses.append(per.detach().clone())
        grads.append(w.grad.detach().clone())
        opt.step()
        traj.append(w.detach().clone())
    save('wplus_256.npz', losses=torch.stack(losses), traj=torch.stack(traj), grads=torch.stack(grads))


def gold_samm():
    """AlignNet / SPM_Warp (2 cycles, with and without a coarser field) at C=8, H=16."""
    from src.ops.SAMM.helpers import SPM_Warp, new_PRM
    C, H, B = 8, 16, 2
    warp = SPM_Warp(C, scale=0.08, cycle_align=2, diff_fAndg=True).eval()
    sd = synth.samm_state(C, 'm', seed=21)
    sub = {k[len('m.alignment.'):]: v for k, v in sd.items() if k.startswith('m.alignment.')}
    print('samm load:', warp.load_state_dict(sub, strict=True))
    src = synth.normal('samm.src', (B, C, H, H), 22)
    tgt = synth.normal('samm.tgt', (B, C, H, H), 22)
    prev = torch.cat([synth.normal('samm.prev.d', (B, 2, H // 2, H // 2), 22, 0.04),
                      synth.uniform('samm.prev.a', (B, 1, H // 2, H // 2), 22)], dim=1)
    g = dict(src=src, tgt=tgt, prev=prev)
    with torch.no_grad():
        g['alignnet'] = warp.body(tgt, src)
        y0, f0 = warp(src, tgt, None, None)
        y1, f1 = warp(src, tgt, None, prev)
        g['warp_out'], g['warp_field'] = y0, f0
        g['warp_out_prev'], g['warp_field_prev'] = y1, f1
        g['prm_up'] = new_PRM(prev[:, 2:], f0[:, 2:])
        g['prm_same'] = new_PRM(f0[:, 2:], f1[:, 2:])
        warp1 = SPM_Warp(C, scale=0.08, cycle_align=1, diff_fAndg=True).eval()
        warp1.load_state_dict(sub, strict=True)
        y2, f2 = warp1(src, tgt, None, prev)
        g['warp1_out_prev'], g['warp1_field_prev'] = y2, f2
    save('samm.npz', **g)


def gold_samm_nodiff():
    """AlignNet / SPM_Warp with diff_fAndg=False (reference SAMM/helpers.py:98-101: the body sees cat([IN(source), IN(target)]) instead of
    cat([IN(source) - IN(target), IN(target)])): the same weights and inputs as gold_samm."""
    from src.ops.SAMM.helpers import SPM_Warp
    C, H, B = 8, 16, 2
    warp = SPM_Warp(C, scale=0.08, cycle_align=2, diff_fAndg=False).eval()
    sd = synth.samm_state(C, 'm', seed=21)
    sub = {k[len('m.alignment.'):]: v for k, v in sd.items() if k.startswith('m.alignment.')}
    print('samm_nodiff load:', warp.load_state_dict(sub, strict=True))
    src = synth.normal('samm.src', (B, C, H, H), 22)
    tgt = synth.normal('samm.tgt', (B, C, H, H), 22)
    prev = torch.cat([synth.normal('samm.prev.d', (B, 2, H // 2, H // 2), 22, 0.04),
                      synth.uniform('samm.prev.a', (B, 1, H // 2, H // 2), 22)], dim=1)
    g = {}
    with torch.no_grad():
        g['alignnet'] = warp.body(tgt, src)
        y1, f1 = warp(src, tgt, None, prev)
        g['warp_out_prev'], g['warp_field_prev'] = y1, f1
    save('samm_nodiff.npz', **g)


def gold_modbtn():
    """The `mod_btn` feature extractors of StyledscaleNshfitBlock (reference src/ops/SAMM/helpers.py:22-57,182-216): the reference
    modules with their own (seeded) initialisation, state dict + input + style + output.  C = 16 -> 16 and 16 -> 32 channels, 24x24."""
    from src.ops.SAMM.helpers import style_bottleneck_IR, styleBlock, StyledscaleNshfitBlock
    g = {}
    B, C, H, SD = 2, 16, 24, 32
    x = synth.normal('modbtn.x', (B, C, H, H), 41)
    style = synth.normal('modbtn.style', (B, SD), 41)
    gen = synth.normal('modbtn.gen', (B, C, H, H), 41)
    g.update(x=x, style=style, gen=gen)
    for tag, mk in (('sb', lambda: style_bottleneck_IR(C, C, SD, bn=False)), ('sb2', lambda: style_bottleneck_IR(C, 2 * C, SD, bn=False)),
                    ('blk', lambda: styleBlock(C, C, SD, noiseInjection=False, activation=False))):
        torch.manual_seed(1234)
        m = mk().eval()
        with torch.no_grad():
            for n_, p_ in m.named_parameters():      # zero biases / unit PReLU slopes would hide a wrong wiring
                if n_.endswith('bias') or 'res_layer.2' in n_:
                    p_.add_(0.1 * torch.randn_like(p_))
            y = m(x, style)
        for k, v in m.state_dict().items():
            g[f'{tag}.sd.{k}'] = v
        g[f'{tag}.y'] = y
    for tag, btn in (('blockA', 'style_bottleneck_IR'), ('blockB', 'styleBlock')):
        torch.manual_seed(4321)
        m = StyledscaleNshfitBlock(C, C, SD, btn=btn, scale=0.08, cycle_align=1, diff_fAndg=True).eval()
        with torch.no_grad():
            y, f = m(x, style, image=gen, aligned=None)
        for k, v in m.state_dict().items():
            g[f'{tag}.sd.{k}'] = v
        g[f'{tag}.y'], g[f'{tag}.field'] = y, f
    save('modbtn.npz', **g)


class _NoiseFeed:
    """Make every NoiseInjection of the reference draw a PRESET noise map instead of a fresh
    RNG sample (the OOD path never exposes ``noise=`` — passing it would bypass the SAMM
    callback, model.py:284-290 — and its RNG stream cannot travel).
    ``Tensor.normal_`` is swapped for the duration of each NoiseInjection.forward only, so
    the reference code itself runs unmodified."""

    def __init__(self, noises):
        self.noises = list(noises)
        self.calls = 0

    def install(self):
        from src.ops.StyleGAN import model as M
        orig = M.NoiseInjection.forward
        feed = self

        def fwd(self_, image, noise=None, **kw):
            if noise is not None:
                return orig(self_, image, noise=noise, **kw)
            preset = feed.noises[feed.calls]
            feed.calls += 1
            real_normal = torch.Tensor.normal_

            def fake_normal(t, *a, **k):
                assert tuple(t.shape) == tuple(preset.shape), (t.shape, preset.shape)
                return t.copy_(preset)

            torch.Tensor.normal_ = fake_normal
            try:
                return orig(self_, image, noise=None, **kw)
            finally:
                torch.Tensor.normal_ = real_normal

        M.NoiseInjection.forward = fwd
        self._restore = lambda: setattr(M.NoiseInjection, 'forward', orig)

    def remove(self):
        self._restore()


def gold_ood(B=1):
    """Full 1024² OOD forward after the encoder.  The e4e encoder is replaced
    by a stand-in that returns recipe tensors; everything downstream is the reference."""
    from src.archs.OOD_faceGAN_e4e_arch import ood_faceGAN_e4e
    m = ood_faceGAN_e4e(out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True,
                        warp_scale=0.08, cycle_align=2, blend_with_gen=True, ModSize=256).eval()
    sd = synth.ood_state(1024, seed=31)
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith('encoder.') for k in res.missing_keys), [k for k in res.missing_keys if not k.startswith('encoder.')][:5]
    enc_lats = synth.make_latents(1024, B, seed=32, std=0.3)
    enc_feats = synth.make_encoder_feats(B, seed=33)

    class FakeEncoder(torch.nn.Module):
        channels = [64, 64, 128, 256, 512]

        def forward(self, x, return_feats=False):
            return enc_lats.clone(), [f.clone() for f in enc_feats] + [None]

    fe = FakeEncoder()
    fe.progressive_stage = m.encoder.progressive_stage
    m.encoder = fe
    x = synth.make_images(1024, B, seed=34)
    noises = synth.make_noises(1024, B, seed=35)
    feed = _NoiseFeed(noises)
    feed.install()
    torch.manual_seed(1234)
    with torch.no_grad():
        out, lats = m(x)
    feed.remove()
    assert feed.calls == 17, feed.calls
    g = dict(out_sub=out[:, :, ::16, ::16], out_crop=out[:, :, 480:544, 480:544], lats=lats,
             out_mean=out.mean(dim=(2, 3)), out_std=out.std(dim=(2, 3)), out_absmax=out.abs().amax())
    for k in (1, 2, 3, 4):
        a = m.aligns[k]
        step = max(1, a.shape[-1] // 32)
        g[f'align{k}_sub'] = a[:, :, ::step, ::step]
        g[f'align{k}_mean'] = a.mean(dim=(2, 3))
    g['align1024_sub'] = m.aligns[1024][:, :, ::16, ::16]
    g['align1024_crop'] = m.aligns[1024][:, :1, 480:544, 480:544]
    # mask strip of run_ood_faceGAN_inversion.py:74-87 (nearest upsample) — integer index work
    sys.argv = ['x']
    masks = []
    for key, val in m.aligns.items():
        masks.append(torch.nn.functional.interpolate(val[:, 2:], size=1024))
    strip = torch.cat(masks, dim=-1)
    g['mask_strip_sub'] = strip[:, :, ::16, ::16]
    g['mask_strip_rows'] = strip[:, :, 500:502, :]
    save('ood_1024.npz', **g)
    return m, x, enc_lats, enc_feats, noises, out


def gold_cond_types(size=32):
    """Generator.forward with conditions on the up-convs reading latents 1 and 3 for cond_type 'SFT' / 'ADD' / 'FUSE'
    (feature_modulation, model.py:558-566,588-610) and StyleGAN2Generator.forward with the same conditions
    (stylegan2_arch.py:583-595).  A fresh copy of the conditions goes into every call ('FUSE' overwrites conditions[0])."""
    from src.ops.StyleGAN.model import Generator
    G = Generator(size, 512, 8).eval()
    G.load_state_dict(synth.generator_state(size, seed=5), strict=True)
    B = 2
    lat = synth.make_latents(size, B, seed=6)
    noises = synth.make_noises(size, B, seed=7)
    conds = lambda: [[synth.normal(f'cond.{k}.0', (B, 512, r, r), 9, 0.5), synth.normal(f'cond.{k}.1', (B, 512, r, r), 10, 0.5)] for k, r in ((0, 8), (1, 16))]
    g = {}
    with torch.no_grad():
        for ct in ('SFT', 'ADD', 'FUSE'):
            img, feat = G(lat, input_is_tensor=True, input_is_latent=True, noise=noises, return_features=True, conditions=conds(),
                          cond_layers=[1, 3], cond_type=ct)
            g[f'image_{ct}'] = img
            g[f'feat_{ct}_sub'] = feat[:, ::16]
        cb = lambda feats, **kw: conds()[kw['index']][0] * 0.25 + kw['style'][:, :1, None, None]
        img, _ = G(lat, input_is_tensor=True, input_is_latent=True, noise=noises, conditions=conds(), cond_layers=[1, 3], cond_type='ADD', callback=cb)
        g['image_ADD_callback'] = img
    save(f'cond_types_s{size}.npz', **g)


def _load_real(name, relpath):
    """import ONE real reference file as module `name` (the package __init__ files pull cv2 / torchvision / lmdb)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join('/root/reference', relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def gold_imgio():
    """Image I/O + PSNR of the harness from the REAL BasicSR functions: ``tensor2img`` / ``img2tensor``
    (BasicSR/basicsr/utils/img_util.py:9-94) and ``calculate_psnr`` (basicsr/metrics/psnr_ssim.py:9-46 with
    metric_util.py / matlab_functions.py).  cv2 is absent from the image: an EMPTY module stands in for the import, and only
    code paths that never call it are exercised (no channel swap: rgb2bgr / bgr2rgb = False; single-channel masks).
    ``calculate_ssim`` (psnr_ssim.py:49-128) calls two cv2 primitives; numpy stand-ins with their documented semantics are put on
    the stub — ``getGaussianKernel(n, sigma)`` = exp(-(i-(n-1)/2)^2 / (2 sigma^2)) normalised to sum 1 as an (n,1) float64 column
    (OpenCV's closed form for n > 7), ``filter2D(src, -1, k)`` = correlation with BORDER_REFLECT_101 (scipy ``mode='mirror'``;
    the reference crops the 5-pixel border anyway) — so everything of the real ``calculate_ssim`` / ``_ssim`` except those two
    primitives is what produces the stored values."""
    from scipy import ndimage

    def _gauss(n, sigma):
        i = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
        k = np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(n, 1)

    _stub('cv2', getGaussianKernel=_gauss, filter2D=lambda src, ddepth, kernel: ndimage.correlate(src, kernel, mode='mirror'))
    sys.modules['torchvision.utils'].make_grid = None
    _load_real('basicsr.utils.matlab_functions', 'BasicSR/basicsr/utils/matlab_functions.py')
    sys.modules['basicsr.utils'].bgr2ycbcr = sys.modules['basicsr.utils.matlab_functions'].bgr2ycbcr
    _stub('basicsr.metrics')
    _load_real('basicsr.metrics.metric_util', 'BasicSR/basicsr/metrics/metric_util.py')
    ps = _load_real('basicsr.metrics.psnr_ssim', 'BasicSR/basicsr/metrics/psnr_ssim.py')
    iu = _load_real('basicsr.utils.img_util', 'BasicSR/basicsr/utils/img_util.py')
    rng = np.random.default_rng(7)
    g = {}
    # values that land exactly on .5 after x255 (numpy rounds half to even), outside the clamp range, and random ones
    k = np.arange(0, 48, dtype=np.float64)
    halves = ((k + 0.5) / 255.0) * 2 - 1
    t = np.concatenate([halves, [-1.3, 1.7, -1.0, 1.0, 0.0], rng.uniform(-1.1, 1.1, 3 * 20 * 21 - 53)]).astype(np.float32).reshape(3, 20, 21)
    t = torch.from_numpy(t)
    g['t'] = t
    g['t2i_rgb_u8'] = iu.tensor2img(t.clone(), rgb2bgr=False, min_max=(-1, 1))
    g['t2i_rgb_f32'] = iu.tensor2img(t.clone(), rgb2bgr=False, out_type=np.float32, min_max=(-1, 1))
    g['t2i_batch1_u8'] = iu.tensor2img(t.clone().unsqueeze(0), rgb2bgr=False, min_max=(-1, 1))
    m = torch.from_numpy(rng.uniform(-0.2, 1.2, (1, 20, 21)).astype(np.float32))
    g['mask'] = m
    g['t2i_mask_u8'] = iu.tensor2img(m.clone(), min_max=(0, 1))              # run_ood_faceGAN_inversion.py:84
    img = rng.integers(0, 256, (20, 21, 3)).astype(np.float64) / 255.0
    g['img_f64'] = img
    g['i2t'] = iu.img2tensor(img.copy(), bgr2rgb=False, float32=True)
    a = rng.integers(0, 256, (24, 28, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.float64) + rng.normal(0, 6, a.shape), 0, 255).round().astype(np.uint8)
    g['psnr_a'], g['psnr_b'] = a, b
    g['psnr_vals'] = np.array([ps.calculate_psnr(a, b, crop_border=0), ps.calculate_psnr(a, b, crop_border=4),
                               ps.calculate_psnr(a, b, crop_border=4, test_y_channel=True),
                               ps.calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), crop_border=2, input_order='CHW'),
                               ps.calculate_psnr(a.astype(np.float64), b.astype(np.float64), crop_border=0, test_y_channel=True)])
    g['ssim_vals'] = np.array([ps.calculate_ssim(a, b, crop_border=0), ps.calculate_ssim(a, b, crop_border=4),
                               ps.calculate_ssim(a, b, crop_border=4, test_y_channel=True),
                               ps.calculate_ssim(a.transpose(2, 0, 1), b.transpose(2, 0, 1), crop_border=2, input_order='CHW'),
                               ps.calculate_ssim(a, a, crop_border=0)])
    save('imgio.npz', **g)


def main():
    install_stubs()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ['ops', 'gen', 'wplus', 'samm', 'ood', 'enc']
    if 'ops' in which:
        gold_ops()
    if 'gen' in which:
        gold_generator(32)
    if 'narrow' in which:
        gold_generator_narrow(64, 0.5)
    if 'narrow_pad' in which:           # channel counts that are not multiples of 16 (the engine zero-pads them): 24 at 256², 8 at 1024²
        gold_generator_narrow(256, 0.1875)
        gold_generator_narrow(1024, 0.25)
    if 'wplus' in which:
        gold_wplus(32, 5)
    if 'wplus1024' in which:
        gold_wplus_1024()
    if 'genb4' in which:
        gold_generator_1024_b4()
    if 'wplus256' in which:
        gold_wplus_256()
    if 'wplus_blur_64' in which:
        gold_wplus_blur(64, (0, 1), 4)
    if 'wplus_blur_256' in which:
        gold_wplus_blur(256, (0, 1), 3, UP_TAPS)
    if 'wplus_blur_1024' in which:
        gold_wplus_blur(1024, (WPLUS_1024_IMAGE,), 2, UP_TAPS)
    if 'resample' in which:
        gold_generator_resample(64)
    for name in WPLUS_LONG:
        if name in which:
            gold_wplus_long(name)
    if 'samm' in which:
        gold_samm()
    if 'samm_nodiff' in which:
        gold_samm_nodiff()
    if 'ood' in which:
        gold_ood(1)
    if 'enc' in which:
        gold_encoder()
    if 'restyle' in which:
        gold_restyle()
    if 'modbtn' in which:
        gold_modbtn()
    if 'fs' in which:
        gold_featurestyle()
    if 'featin' in which:
        gold_features_in()
    if 'imgio' in which:
        gold_imgio()
    if 'cond' in which:
        gold_cond_types()
    if 'c1' in which:
        gold_cli_c1()


def gold_cli_c1():
    """BASELINE configs[0] / SURVEY §8f N2: what the reference CLI WRITES for one 256x256 image, as uint8.
    The per-image body of run_ood_faceGAN_inversion.py:159-180 on the real classes: ``cv2.imread(f) / 255.0`` ->
    ``img2tensor`` (real, img_util.py:9-37) -> ``(t - 0.5) * 2`` -> ``F.interpolate(bilinear)`` to 1024² (:162-163) ->
    the reference ``ood_faceGAN_e4e`` INCLUDING its e4e encoder (recipe weights, preset noise maps) -> real ``tensor2img``
    (img_util.py:40-94: clamp, [0,1], x255, ``.round()``, uint8) -> ``extract_masks`` (:74-87: nearest to 1024, side by
    side, real ``tensor2img``).  cv2 is absent: the run script itself cannot be imported, so its ten lines are spelled out
    here around the real functions; the BGR<->RGB swap (``cv2.cvtColor``, a channel reversal) is done with numpy and the
    real functions are called with ``bgr2rgb`` / ``rgb2bgr`` = False.  The input image is stored; of the 1024² uint8
    outputs a 4x sub-sampling and a full-resolution crop are stored, the masks at their native resolutions (the strip is
    their nearest up-sampling: every strip pixel can be rebuilt from them)."""
    import torch.nn.functional as F
    from src.archs.OOD_faceGAN_e4e_arch import ood_faceGAN_e4e
    _stub('cv2')
    sys.modules['torchvision.utils'].make_grid = None
    _load_real('basicsr.utils.matlab_functions', 'BasicSR/basicsr/utils/matlab_functions.py')
    sys.modules['basicsr.utils'].bgr2ycbcr = sys.modules['basicsr.utils.matlab_functions'].bgr2ycbcr
    _stub('basicsr.metrics')
    _load_real('basicsr.metrics.metric_util', 'BasicSR/basicsr/metrics/metric_util.py')
    ps = _load_real('basicsr.metrics.psnr_ssim', 'BasicSR/basicsr/metrics/psnr_ssim.py')
    iu = _load_real('basicsr.utils.img_util', 'BasicSR/basicsr/utils/img_util.py')

    m = ood_faceGAN_e4e(out_size=1024, style_dim=512, encoder='E4E', enable_modulation=True,
                        warp_scale=0.08, cycle_align=2, blend_with_gen=True, ModSize=256).eval()
    sd = synth.ood_state(1024, seed=31)
    shapes = {k: tuple(v.shape) for k, v in m.encoder.state_dict().items()}
    enc = synth.encoder_state(shapes, seed=41)
    sd.update({'encoder.' + k: (v * 0.1 if k.endswith('linear.weight') else v) for k, v in enc.items()})
    print('c1 load:', m.load_state_dict(sd, strict=True))
    m.delta_latent.data = torch.zeros_like(m.delta_latent)                    # load_model, run_ood_faceGAN_inversion.py:45

    # a smooth random field plus pixel noise, 256x256 BGR uint8 ("single 256x256 random face")
    rng = np.random.default_rng(71)
    low = rng.uniform(0, 255, (3, 8, 8))
    low = F.interpolate(torch.from_numpy(low)[None], size=(256, 256), mode='bicubic', align_corners=False)[0].numpy()
    bgr = np.clip(low.transpose(1, 2, 0) + rng.normal(0, 20, (256, 256, 3)), 0, 255).round().astype(np.uint8)

    cv2im = bgr / 255.0                                                      # :160
    rgb = np.ascontiguousarray(cv2im[:, :, ::-1])                            # cv2.cvtColor(BGR2RGB) inside img2tensor
    input_im = (torch.stack(iu.img2tensor([rgb], bgr2rgb=False), dim=0) - 0.5) * 2
    assert input_im.dtype == torch.float32
    if input_im.shape[-1] != 1024:
        input_im = F.interpolate(input_im, size=(1024, 1024), mode='bilinear')
    noises = synth.make_noises(1024, 1, seed=35)
    feed = _NoiseFeed(noises)
    feed.install()
    torch.manual_seed(1234)
    with torch.no_grad():
        inversion_im, lats = m(input_im)
    feed.remove()
    assert feed.calls == 17, feed.calls
    out_f32_sub = inversion_im[0, :, ::16, ::16].clone()                      # tensor2img clamps its argument IN PLACE
    res_rgb = iu.tensor2img(inversion_im, rgb2bgr=False, min_max=(-1, 1))     # save_img_to :64-72
    result = np.ascontiguousarray(res_rgb[:, :, ::-1])                        # cv2.cvtColor(RGB2BGR) inside tensor2img
    assert result.dtype == np.uint8 and result.shape == (1024, 1024, 3)
    masks = []
    for key_ in sorted(m.aligns.keys()):                                      # extract_masks :74-87
        mask = m.aligns[key_][:, 2:, ...]
        masks.append(F.interpolate(mask, size=(1024, 1024)))
    strip = iu.tensor2img(torch.cat(masks, dim=3)[0, ...], min_max=(0, 1))
    assert strip.dtype == np.uint8 and strip.shape == (1024, 5 * 1024)
    g = dict(bgr=bgr, out_u8_sub=result[::4, ::4], out_u8_crop=result[448:576, 448:576], lats=lats,
             out_f32_sub=out_f32_sub, n_saturated=np.int64(((result == 0) | (result == 255)).sum()))
    for i, s_ in enumerate((32, 64, 128, 256)):
        st = 1024 // s_
        native = strip[::st, 1024 * i:1024 * (i + 1):st]
        assert np.array_equal(np.repeat(np.repeat(native, st, 0), st, 1), strip[:, 1024 * i:1024 * (i + 1)])
        g[f'mask{i + 1}_u8'] = native
    g['mask1024_u8_sub'] = strip[::4, 4096::4]
    g['mask1024_u8_crop'] = strip[448:576, 4096 + 448:4096 + 576]
    # metrics as the harness evaluates them when the input is not 1024² (the reference asserts equal shapes: its gt is then
    # the resized input): PSNR from the real function
    gt = np.ascontiguousarray(iu.tensor2img(input_im, rgb2bgr=False, min_max=(-1, 1))[:, :, ::-1])
    g['psnr_resized_gt'] = np.float64(ps.calculate_psnr(gt.astype(np.float64), result, crop_border=2, test_y_channel=False))
    print('c1: saturated', int(g['n_saturated']), 'of', result.size, 'psnr', float(g['psnr_resized_gt']))
    save('cli_c1.npz', **g)


def gold_encoder():
    """e4e encoder (IR-SE-50 + FPN + 18 GradualStyleBlocks) at 256², B=1, eval mode, recipe weights."""
    from src.ops.e4e.encoders.psp_encoders import Encoder4Editing

    class O(dict):
        __getattr__ = dict.__getitem__
    enc = Encoder4Editing(50, 'ir_se', O(stylegan_size=1024), bn=True).eval()
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    print('encoder load:', enc.load_state_dict(synth.encoder_state(shapes, seed=41), strict=True))
    x = synth.make_images(256, 1, seed=42)
    with torch.no_grad():
        w, feats = enc(x, return_feats=True)
    g = dict(w=w)
    for i, f in enumerate(feats):
        step = max(1, f.shape[-1] // 16)
        g[f'feat{i}_sub'] = f[:, ::8, ::step, ::step]
        g[f'feat{i}_mean'] = f.mean(dim=(2, 3))
    save('encoder_256.npz', **g)


def gold_restyle(B=1):
    """ReStyle variant: full forward of ood_faceGAN_restyle at 1024², enc_cycle=2 — average image,
    two encoder passes, one plain reconstruction, then the OOD forward.  Recipe weights; preset noise for the three
    generator passes (avg image at batch 1, cycle reconstruction, final)."""
    import tempfile
    from src.archs.OOD_faceGAN_restyle_arch import ood_faceGAN_restyle
    ck = synth.restyle_checkpoint(seed=51)
    with tempfile.TemporaryDirectory() as d:
        pth = os.path.join(d, 'restyle.pth')
        torch.save(ck, pth)
        m = ood_faceGAN_restyle(out_size=1024, style_dim=512, encoder='ReStyle', ReStyle_pth=pth, enc_cycle=2,
                                enable_modulation=True, warp_scale=0.08, cycle_align=2, blend_with_gen=True, ModSize=256).eval()
    sd = synth.ood_state(1024, seed=31)
    sd.pop('avg_latent')
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith('encoder.') or k == 'avg_latent' for k in res.missing_keys), res.missing_keys[:5]
    x = synth.make_images(1024, B, seed=52)
    passes = [synth.make_noises(1024, 1, seed=53), synth.make_noises(1024, B, seed=54), synth.make_noises(1024, B, seed=55)]
    feed = _NoiseFeed([n for p in passes for n in p])
    feed.install()
    with torch.no_grad():
        out, lats = m(x)
    feed.remove()
    assert feed.calls == 51, feed.calls
    g = dict(out_sub=out[:, :, ::16, ::16], out_crop=out[:, :, 480:544, 480:544], lats=lats, avg_img_sub=m.avg_img[:, :, ::4, ::4],
             out_mean=out.mean(dim=(2, 3)), out_std=out.std(dim=(2, 3)))
    for k in (1, 2, 3, 4):
        a = m.aligns[k]
        step = max(1, a.shape[-1] // 32)
        g[f'align{k}_sub'] = a[:, :, ::step, ::step]
        g[f'align{k}_mean'] = a.mean(dim=(2, 3))
    g['align1024_sub'] = m.aligns[1024][:, :1, ::16, ::16]
    save('restyle_1024.npz', **g)


def gold_featurestyle(B=1):
    """Feature-Style variant: full forward of ood_faceGAN_FeatureStyle at 1024², recipe weights, preset
    noise.  The reference builds its trunk from an ArcFace IResNet-50 checkpoint before FeatureStyle_pth overwrites it:
    a freshly initialised iresnet50 state stands in for that file."""
    import tempfile
    from src.archs.OOD_faceGAN_featureStyle_arch import ood_faceGAN_FeatureStyle
    from src.ops.FeatureStyle.arcface.iresnet import iresnet50
    with tempfile.TemporaryDirectory() as d:
        arc, pth, avg = (os.path.join(d, n) for n in ('arc.pth', 'fs.pth', 'avg.pth'))
        torch.save(iresnet50().state_dict(), arc)
        torch.save(synth.featurestyle_state(seed=61), pth)
        torch.save(synth.normal('fs.latent_avg', (18, 512), 61, 0.5), avg)
        m = ood_faceGAN_FeatureStyle(out_size=1024, style_dim=512, encoder='FeatureStyle', FeatureStyle_pth=pth, arcface_model_path=arc,
                                     avg_latent_pth=avg, enable_modulation=True, warp_scale=0.08, cycle_align=2, blend_with_gen=True,
                                     ModSize=256).eval()
    sd = synth.ood_state(1024, seed=31)
    sd.pop('avg_latent')
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith('encoder.') or k == 'avg_latent' for k in res.missing_keys), res.missing_keys[:5]
    x = synth.make_images(1024, B, seed=62)
    feed = _NoiseFeed(synth.make_noises(1024, B, seed=63))
    feed.install()
    with torch.no_grad():
        out, lats = m(x)
        _, content, taps = m.encoder(m.face_pool(x), return_feats=True)
    feed.remove()
    assert feed.calls == 17, feed.calls
    g = dict(out_sub=out[:, :, ::16, ::16], out_crop=out[:, :, 480:544, 480:544], lats=lats, content_sub=content[:, ::8],
             out_mean=out.mean(dim=(2, 3)), out_std=out.std(dim=(2, 3)))
    for i, f in enumerate(taps):
        g[f'tap{i}_mean'] = f.mean(dim=(2, 3))
    for k in (1, 2, 3, 4):
        a = m.aligns[k]
        step = max(1, a.shape[-1] // 32)
        g[f'align{k}_sub'] = a[:, :, ::step, ::step]
    g['align1024_sub'] = m.aligns[1024][:, :1, ::16, ::16]
    save('featurestyle_1024.npz', **g)


def gold_features_in(size=32):
    """`insert_feature` of Generator.forward (model.py:541-546,557,572): features mixed into the inputs of the styled
    convs reading latents 4 (plain conv at 16²) and 5 (up-conv 16² -> 32²), feature_scale 0.3 and 1.0."""
    from src.ops.StyleGAN.model import Generator
    G = Generator(size, 512, 8).eval()
    G.load_state_dict(synth.generator_state(size, seed=5), strict=True)
    B = 2
    lat = synth.make_latents(size, B, seed=6)
    noises = synth.make_noises(size, B, seed=7)
    f4 = synth.normal('featin.4', (B, 512, 16, 16), 9)
    f5 = synth.normal('featin.5', (B, 512, 16, 16), 10)
    feats = [None] * 8
    feats[4], feats[5] = f4, f5
    g = {}
    with torch.no_grad():
        for fs in (0.3, 1.0):
            img, feat = G(lat, input_is_tensor=True, input_is_latent=True, noise=noises, return_features=True, features_in=feats,
                          feature_scale=fs)
            g[f'image_fs{fs}'] = img
            g[f'feat_fs{fs}_sub'] = feat[:, ::16]
    save(f'features_in_s{size}.npz', **g)


if __name__ == '__main__':
    main()
