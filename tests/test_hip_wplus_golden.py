"""W+ step pinned at the BENCHMARKED geometry (BASELINE configs[2]: 1024², the bench recipe's weights and inputs) against
vectors produced by the reference Generator's own autograd (tests/golden/make_golden.py: gold_wplus_1024 / gold_wplus_256),
plus the forward range control of the split-f16 path."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oodgan import synth  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _recipe(size, gidx, dev):
    """bench.py's synthetic inputs for the global image indices ``gidx``."""
    cat = lambda parts: torch.cat(parts, 0).to(dev)
    target = cat([synth.make_images(size, 1, seed=1000 + g) for g in gidx])
    w0 = cat([synth.make_latents(size, 1, seed=3000 + g, std=0.3) for g in gidx])
    per = [synth.make_noises(size, 1, seed=2000 + g) for g in gidx]
    noises = [cat([n[i] for n in per]) for i in range(len(per[0]))]
    return target, w0, noises


@pytest.mark.parametrize('prec', ['f16s', 'f32', 'f16s-g2'])
@pytest.mark.parametrize('batch', ['alone', 'image3of8'])
def test_wplus_step_1024_vs_reference_autograd(dev, golden, prec, batch):
    """loss, image and dL/dW+ of one W+ step at 1024² vs the reference Generator evaluated in float64; every production
    kernel instance of the bench (strip kernel, 16x32-tile kernel at 512², stride-2 / transposed kernels at 512²<->1024²,
    fused producers) is on the checked path.  'image3of8': the golden image sits at index 3 of a batch of 8, and the
    In both cases the SECOND step of the loop is checked too (carried range scales, fused producers): with the split-f16
    arithmetic that step must keep the 1024² activations in F-form and run conv_f16s_stripx (forward + input gradient) —
    asserted through the layout of the saved activations and the library's dispatch counters, so that a changed eligibility
    rule cannot silently move this test back onto the S-form strip kernel."""
    from oodgan.engine import GeneratorEngine
    from oodgan import ops, _lib
    g = golden('wplus_1024.npz')
    size, gi = 1024, int(g['image_index'])
    gidx = [gi] if batch == 'alone' else [0, 1, 2, gi, 4, 5, 6, 7]
    k = gidx.index(gi)
    target, w0, noises = _recipe(size, gidx, dev)
    eng = GeneratorEngine({n: v.to(dev) for n, v in synth.generator_state(size, seed=0).items()}, size, precision=prec)
    gmul = ops.loss_scale_for(3 * size * size)
    gref = g['grad_f64'][0]
    eng.reset_bwd_state()
    eng.reset_fwd_state()
    for rep in range(2):
        # rep 0: exact range scales (first step of the loop); rep 1: the same latents again through the carried-scale
        # forward and the fused backward producers — must reproduce the same gradient
        _lib.dispatch_reset()
        img = eng.forward(w0, noises, save=True, range_mode='carry')
        saved_acts = dict(eng.saved['acts'])
        fform = [n for n in ('convs.14', 'convs.15') if isinstance(saved_acts[n], ops.FForm)]
        loss, gimg = ops.mse_loss_grad(img, target, gmul)
        glat = eng.backward(gimg, gmul, carry_scale=True)
        nx, ns = _lib.dispatch_count('stripx'), _lib.dispatch_count('strip')
        g2 = {k: _lib.dispatch_count(k) for k in ('s1big_g2', 's2big_g2', 'stripx_g2')}
        if prec == 'f16s-g2':
            # round 6: every input-gradient conv of the 8-wave / strip families ran its two-instruction instance (the gradient operand rounded to f16)
            assert g2['s1big_g2'] >= 3 and g2['s2big_g2'] >= 3 and g2['stripx_g2'] == (1 if rep == 1 else 0), g2
            # ... and in the steady state the up-conv layers' gradients travel as 32-byte hi-only records (blur^T strip producer -> stride-2 conv)
            nxh = _lib.dispatch_count('s2big_xh')
            assert (nxh == 0) if rep == 0 else (3 <= nxh <= 5), nxh
            n1xh = _lib.dispatch_count('s1big_xh')      # ... and the conv layers' (stride-2 conv's fused epilogue -> stride-1 input-gradient conv)
            assert (n1xh == 0) if rep == 0 else (3 <= n1xh <= 5), n1xh
        else:
            assert not any(g2.values()), g2
        if prec in ('f16s', 'f16s-g2') and rep == 1:
            # steady state of the loop: up-conv tail and last conv in F-form, both 1024² convs in conv_f16s_stripx
            assert fform == ['convs.14', 'convs.15'], fform
            assert (nx, ns) == (2, 0), (nx, ns)
            assert _lib.dispatch_count('upvb') == 1          # ... and its up-conv in one pass (conv_f16s_upvb.hip)
            # round 4's fused paths BY NAME: the conv layers of the 64² ... 512² levels (convs.7 / 9 / 11 / 13) hand ToRGB sums and the next
            # up-conv's S-form out of the 8-wave conv's epilogue, their activation is saved only as that S-form, and the fused epilogue of
            # the stride-2 conv above decodes it — a changed eligibility rule (bwd_state, s1_ys_supported, s2_fuse_supported) must fail here
            # instead of silently moving this test back onto the two-pass path
            sf = [n for n in ('convs.7', 'convs.9', 'convs.11', 'convs.13') if isinstance(saved_acts[n], ops.SFormSaved)]
            # at batch 8 all four levels; one image alone fills the 8-wave kernel's grid (>= 128 work items) from the 128² level on
            want = ['convs.7', 'convs.9', 'convs.11', 'convs.13'] if len(gidx) == 8 else ['convs.9', 'convs.11', 'convs.13']
            assert sf == want, sf
            assert _lib.dispatch_count('s1big_ys') == len(want), _lib.dispatch_count('s1big_ys')
            assert _lib.dispatch_count('s2big_dotx_sform') == len(want), _lib.dispatch_count('s2big_dotx_sform')
            assert _lib.dispatch_count('s2big_fuse') >= len(want)
        elif prec in ('f16s', 'f16s-g2'):
            assert fform == [] and (nx, ns) == (0, 2), (fform, nx, ns)      # first step: S-form strip kernel, exact scales
        else:
            assert fform == [] and nx == 0
        im = img[k:k + 1].double().cpu()
        e_img = max((im[:, :, ::16, ::16] - g['image_sub']).abs().max().item(), (im[:, :, 480:544, 480:544] - g['image_crop']).abs().max().item())
        e_mom = max((im.mean(dim=(2, 3)) - g['image_mean']).abs().max().item(), (im.std(dim=(2, 3)) - g['image_std']).abs().max().item())
        e_loss = abs(loss[k].item() - g['loss_f64'].item()) / g['loss_f64'].item()
        rel = (glat[k].double().cpu() - gref).abs().max().item() / gref.abs().max().item()
        print(f'[{prec} {batch} rep{rep}] 1024² W+ step vs reference f64: |d image| {e_img:.2e} (absmax {g["image_absmax"].item():.2f}), '
              f'moments {e_mom:.2e}, loss rel {e_loss:.2e}, dL/dw rel {rel:.2e}')
        assert e_img < 1e-3 and e_mom < 1e-5 and e_loss < 1e-5
        # 'f16s-g2': the gradient operand of every contraction is rounded to f16 (2^-11, zero mean): bar 3e-4
        assert rel < (3e-4 if prec == 'f16s-g2' else 1e-4), rel
    assert not eng.bwd_scale_violated() and not eng.fwd_range_violated()


def test_wplus_trajectory_256_vs_golden(dev, golden):
    """5 Adam steps at 256² (B=2) vs the trajectory of the reference Generator + torch.optim.Adam."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    g = golden('wplus_256.npz')
    size, B = 256, 2
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size)
    target = synth.make_images(size, B, seed=71).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=72)]
    w0 = synth.make_latents(size, B, seed=73, std=0.3).to(dev)
    w, losses, traj = WPlusInverter(eng).invert(target, w0, noises, steps=5, return_trajectory=True)
    el = (losses.cpu() - g['losses']).abs().max().item() / g['losses'].abs().max().item()
    dw = (torch.stack(traj).cpu() - g['traj']).abs()
    print(f'256² trajectory: loss rel err {el:.2e}, max |dw| {dw.max().item():.2e}, within 2e-3: {(dw < 2e-3).float().mean().item():.5f}')
    assert el < 1e-4
    assert (dw < 2e-3).float().mean().item() > 0.999
    assert (losses[-1] < losses[0]).all()


def _hot_state(size, seed, gain_input, gain_style):
    """A generator whose activations leave the f16 range: the constant input is scaled (activations scale with it, the
    demodulation only normalises the weights) and one modulation bias is scaled (style of that layer)."""
    P = synth.generator_state(size, seed=seed)
    P['input.input'] = P['input.input'] * gain_input
    P['convs.3.conv.modulation.bias'] = P['convs.3.conv.modulation.bias'] * gain_style
    return P


@pytest.mark.parametrize('gain_input,gain_style', [(3.0e4, 1.0), (2.0e3, 300.0), (1.0e-6, 1.0)])
def test_forward_range_guard_large_activations(dev, gain_input, gain_style):
    """x*s far above 65504 (or far below the f16 normal range) in the S-form producers: the per-layer power-of-two range
    scale keeps the split-f16 path at fp32-class accuracy.  Ground truth: the oracle in float64 (forward and autograd)."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    from oodgan import ops
    size, B = 64, 2
    P = _hot_state(size, 5, gain_input, gain_style)
    lat = synth.make_latents(size, B, seed=14)
    noises = synth.make_noises(size, B, seed=7)
    target = synth.make_images(size, B, seed=9)
    w = lat.double().requires_grad_(True)
    img_ref, feats = R.generator_forward({k: v.double() for k, v in P.items()}, w, [n.double() for n in noises], size, return_features=True)
    R.wplus_loss(img_ref, target.double()).backward()
    amax = max(f.abs().max().item() for f in feats)
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
    img = eng.forward(lat.to(dev), [n.to(dev) for n in noises], save=True)
    assert torch.isfinite(img).all()
    scale = img_ref.abs().max().item()
    e_img = (img.double().cpu() - img_ref.detach()).abs().max().item() / scale
    gmul = ops.loss_scale_for(3 * size * size)
    loss, gimg = ops.mse_loss_grad(img, target.to(dev), gmul)
    glat = eng.backward(gimg, gmul)
    assert torch.isfinite(glat).all()
    rel = (glat.double().cpu() - w.grad).abs().max().item() / w.grad.abs().max().item()
    print(f'range guard gain ({gain_input:g},{gain_style:g}): max|activation| {amax:.3g}, image rel err {e_img:.2e}, dL/dw rel err {rel:.2e}')
    assert e_img < 1e-4 and rel < 1e-3
    # the W+ loop (carried scales, fused producers) on the same generator
    wl, losses = WPlusInverter(eng).invert(target.to(dev), lat.to(dev), [n.to(dev) for n in noises], steps=3)
    w_ref, l_ref = R.wplus_invert({k: v.double() for k, v in P.items()}, target.double(), lat.double(), [n.double() for n in noises], size, steps=3)
    assert torch.isfinite(wl).all()
    assert (losses.double().cpu() - l_ref).abs().max().item() <= 1e-3 * l_ref.abs().max().item()
    assert not eng.fwd_range_violated()


def test_forward_range_violation_falls_back_to_exact(dev):
    """A carried forward scale that no longer fits (sabotaged after the first step) raises the device flag and the loop is
    re-run with exact per-step scales."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, B = 32, 2
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=5).items()}, size)
    target = synth.make_images(size, B, seed=9).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    w0 = synth.make_latents(size, B, seed=14).to(dev)
    inv = WPlusInverter(eng, use_plan=False)      # the fault is injected through the Python-driven step (call counting)
    w_ref, l_ref = inv.invert(target, w0, noises, steps=4)
    orig, calls = eng.forward, {'n': 0, 'carry': 0}

    def sabotaged(*a, **kw):
        r = orig(*a, **kw)
        calls['n'] += 1
        if calls['n'] == 1 and eng.carry_range:
            eng.fwd_range.q[3].mul_(2.0 ** 12)          # x*s*q of layer 3 now reaches ~4e6: outside [2^-8, 2^15)
        return r
    eng.forward = sabotaged
    w, l = inv.invert(target, w0, noises, steps=4)
    del eng.forward
    assert calls['n'] == 8                               # 4 flagged steps + the 4 steps of the exact re-run
    assert torch.isfinite(w).all() and (l - l_ref).abs().max().item() <= 1e-5 * l_ref.abs().max().item()
    assert eng.carry_range and eng.fused_bwd
