"""``Generator(blur_kernel=<four taps>)`` (reference model.py:376-384; VERDICT r5 "missing" item 3): every Blur / Upsample of the chain — the up-conv
tails (separate pass, strip walk, one-pass 1024² kernel), ToRGB's skip up-sampling, and their adjoints in the W+ backward — with an ASYMMETRIC filter,
[1,4,2,1], against the reference Generator's own autograd in float64 (tests/golden/make_golden.py: gold_wplus_blur).  With the shipped [1,3,3,1] a
flipped, transposed or mirrored kernel in a fused producer is invisible; here it is not."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oodgan import synth  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _inputs(g, size, dev):
    gidx = [int(i) for i in g['image_indices']]
    cat = lambda parts: torch.cat(parts, 0).to(dev)
    target = cat([synth.make_images(size, 1, seed=1000 + i) for i in gidx])
    per = [synth.make_noises(size, 1, seed=2000 + i) for i in gidx]
    noises = [cat([n[k] for n in per]) for k in range(len(per[0]))]
    w0 = cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in gidx])
    return target, w0, noises


@pytest.mark.parametrize('prec', ['f16s-g2', 'f32'])
@pytest.mark.parametrize('size', [64, 256, 1024])
def test_wplus_with_an_asymmetric_blur_kernel_vs_reference(dev, golden, size, prec):
    from oodgan.engine import GeneratorEngine, WPlusInverter
    from oodgan import ops
    if size == 1024 and prec == 'f32':
        pytest.skip('the 1024² case pins the one-pass / strip kernels of the split-f16 path')
    g = golden(f'wplus_blur_{size}.npz')
    taps, up_taps = tuple(int(t) for t in g['taps']), tuple(int(t) for t in g['up_taps'])
    assert taps != taps[::-1] and (size == 64 or up_taps != up_taps[::-1])
    target, w0, noises = _inputs(g, size, dev)
    state = {k: v.to(dev) for k, v in synth.generator_state(size, seed=0, blur_kernel=taps, upsample_kernel=up_taps).items()}
    eng = GeneratorEngine(state, size, precision=prec)
    assert torch.allclose(eng.k4x4.cpu(), synth.make_kernel(taps) * 4.0) and torch.allclose(eng.k_up.cpu(), synth.make_kernel(up_taps) * 4.0)
    st = max(size // 64, 1)
    gmul = ops.loss_scale_for(3 * size * size)
    eng.reset_bwd_state()
    eng.reset_fwd_state()
    gref = g['grad_f64']
    for rep in range(2):            # exact scales, then the carried-scale step with the fused producers (the loop's steady state)
        img = eng.forward(w0, noises, save=True, range_mode='carry')
        loss, gimg = ops.mse_loss_grad(img, target, gmul)
        glat = eng.backward(gimg, gmul, carry_scale=True)
        e_img = (img.double().cpu()[:, :, ::st, ::st] - g['image_sub']).abs().max().item()
        e_loss = ((loss.double().cpu() - g['losses'][0]).abs() / g['losses'][0]).max().item()
        rel = (glat.double().cpu() - gref).abs().max().item() / gref.abs().max().item()
        print(f'[{size} {prec} rep{rep}] blur {taps}: |d image| {e_img:.2e} (absmax {g["image_absmax"].item():.2f}), loss rel {e_loss:.2e}, dL/dw rel {rel:.2e}')
        assert e_img < 1e-3 and e_loss < 1e-5
        assert rel < (3e-4 if prec == 'f16s-g2' else 1e-4), rel
    steps = g['losses'].shape[0]
    _, losses, traj = WPlusInverter(eng).invert(target, w0, noises, steps=steps, return_trajectory=True)
    el = ((losses.double().cpu() - g['losses']).abs() / g['losses']).max().item()
    dw = (torch.stack(traj).double().cpu() - g['traj']).abs()
    print(f'[{size} {prec}] {steps} Adam steps: loss rel {el:.2e}, max |dw| {dw.max().item():.2e}, within 2e-3: {(dw < 2e-3).float().mean().item():.5f}')
    # 'f32' (exact fp32 MFMA, fmaf order differs from the f64 reference): an Adam step amplifies an fp32-level gradient difference on ~0 coordinates
    assert el < (5e-4 if prec == 'f32' else 1e-4)
    assert (dw < 2e-3).float().mean().item() > 0.999
    assert not eng.bwd_scale_violated() and not eng.fwd_range_violated()
    # the default kernel must not reproduce the fixture
    eng0 = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}, size, precision=prec, with_backward=False)
    img0 = eng0.forward(w0, noises)
    assert (img0.double().cpu()[:, :, ::st, ::st] - g['image_sub']).abs().max().item() > 1e-2


def test_generator_module_with_blur_kernel_argument(dev, golden):
    """the module surface: ``Generator(size, 512, 8, blur_kernel=[1,4,2,1])`` builds its Blur buffers from the taps and — like model.py:455 — ToRGB's
    Upsample from [1,3,3,1]; the forward matches the reference module built the same way.  ``StyleGAN2Generator(resample_kernel=...)`` passes the
    taps to both families (stylegan2_arch.py:455-494).  Kernels of another length and states whose layers disagree are refused loudly."""
    from oodgan.modules import Generator, StyleGAN2Generator
    from oodgan.engine import GeneratorEngine
    g = golden('wplus_blur_64.npz')
    size, taps = 64, [int(t) for t in g['taps']]
    target, w0, noises = _inputs(g, size, dev)
    G = Generator(size, 512, 8, blur_kernel=taps).to(dev).eval()
    sd = synth.generator_state(size, seed=0, blur_kernel=tuple(taps))
    # the constructor's buffers already are what the reference constructor builds; everything else comes from the state
    assert all(torch.equal(G.state_dict()[k].cpu(), sd[k]) for k in sd if k.endswith('.kernel'))
    assert not torch.equal(sd['convs.0.conv.blur.kernel'], sd['to_rgbs.0.upsample.kernel'])
    G.load_state_dict({k: v for k, v in sd.items() if not k.endswith('.kernel')}, strict=False)
    with torch.no_grad():
        img, _ = G(w0, input_is_tensor=True, input_is_latent=True, noise=noises)
    e = (img.double().cpu() - g['image_sub']).abs().max().item()
    print(f'Generator(64, blur_kernel={taps}) vs reference: {e:.2e}')
    assert e < 1e-3
    # BasicSR-layout wrapper: resample_kernel reaches the same engine
    gr = golden('generator_resample_s64.npz')
    S = StyleGAN2Generator(size, resample_kernel=tuple(int(t) for t in gr['taps'])).to(dev).eval()
    ros = synth.generator_state(size, seed=5)
    S.load_state_dict({S._ros_to_basicsr(k): v for k, v in ros.items() if not k.endswith('.kernel')}, strict=True)
    with torch.no_grad():
        img2, _ = S(synth.make_latents(size, 2, seed=6).to(dev), input_is_latent=True, noise=[n.to(dev) for n in synth.make_noises(size, 2, seed=7)])
    e2 = (img2.cpu() - gr['image']).abs().max().item()
    print(f'StyleGAN2Generator(64, resample_kernel={gr["taps"].tolist()}) vs reference: {e2:.2e}')
    assert e2 < 1e-3
    with pytest.raises(NotImplementedError):
        Generator(size, 512, 8, blur_kernel=[1, 2, 1])
    bad = {k: v.to(dev) for k, v in sd.items()}
    bad['convs.2.conv.blur.kernel'] = bad['convs.2.conv.blur.kernel'].flip(0)
    with pytest.raises(NotImplementedError):
        GeneratorEngine(bad, size)
