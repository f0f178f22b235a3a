"""GPU parity of the whole generator path: forward vs golden vectors / oracle, backward w.r.t. W+
latents vs torch autograd through the oracle, W+ Adam trajectory vs the golden trajectory that was
produced by the reference Generator."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oodgan import synth  # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def maxdiff(a, b):
    return (a.detach().cpu() - b).abs().max().item()


def test_generator_module_s32_vs_golden(dev, golden):
    from oodgan.modules import Generator
    g = golden('generator_s32.npz')
    G = Generator(32, 512, 8)
    G.load_state_dict(synth.generator_state(32, seed=5), strict=True)
    G = G.to(dev).eval()
    lat = synth.make_latents(32, 2, seed=6).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(32, 2, seed=7)]
    img, feat = G(lat, input_is_tensor=True, input_is_latent=True, noise=noises, return_features=True)
    assert maxdiff(img, g['image']) < 1e-3
    assert maxdiff(feat[:, ::16], g['last_feature_sub']) < 1e-3
    img2, lat2 = G([g['z'].to(dev)], randomize_noise=False, return_latents=True)
    assert maxdiff(lat2, g['latent_from_z']) < 1e-3
    assert maxdiff(img2, g['image_from_z']) < 1e-3
    img3, _ = G([g['z'].to(dev)], randomize_noise=False, truncation=0.7, truncation_latent=g['mean_lat'].to(dev))
    assert maxdiff(img3, g['image_trunc']) < 1e-3


def test_stylegan2generator_basicsr_keys(dev, golden):
    from oodgan.modules import StyleGAN2Generator, basicsr_to_rosinality_key
    g = golden('generator_s32.npz')
    ros = synth.generator_state(32, seed=5)
    G = StyleGAN2Generator(32)
    bsd = {G._ros_to_basicsr(k): v for k, v in ros.items() if not k.endswith('.kernel')}
    assert all(basicsr_to_rosinality_key(bk) in ros for bk in bsd)
    G.load_state_dict(bsd, strict=True)
    G = G.to(dev)
    lat = synth.make_latents(32, 2, seed=6).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(32, 2, seed=7)]
    img, _ = G(lat, input_is_latent=True, noise=noises)
    assert maxdiff(img, g['image']) < 1e-3
    assert set(G.state_dict().keys()) == set(bsd.keys())


@pytest.mark.parametrize('size,narrow', [(64, 0.5), (256, 0.1875), (1024, 0.25)])
def test_stylegan2generator_narrow_vs_reference(dev, golden, size, narrow):
    """``StyleGAN2Generator(out_size, narrow=...)`` (stylegan2_arch.py:422,435-443: every channel count x narrow) against the reference module's
    output (tests/golden/make_golden.py: gold_generator_narrow).  64² / 0.5: 256 channels throughout.  256² / 0.1875: 96 ... 48, 24 channels and
    1024² / 0.25: 128 ... 16, 8 channels — counts that are not multiples of the matrix kernels' 16-channel block: the engine runs the same function on
    zero-padded tensors (engine._pad_channels_to_16); the W+ gradient of such a generator is checked against the oracle's float64 autograd."""
    from oodgan.modules import StyleGAN2Generator
    from oodgan.engine import GeneratorEngine
    from oodgan import ops
    g = golden(f'generator_narrow_s{size}.npz')
    G = StyleGAN2Generator(size, narrow=narrow)
    ros = synth.generator_state(size, seed=5, narrow=narrow)
    log = size.bit_length() - 1
    assert [G._inner[0].channels[2 ** i] for i in range(2, log + 1)] == [int(c) for c in g['channels']]
    G.load_state_dict({G._ros_to_basicsr(k): v for k, v in ros.items() if not k.endswith('.kernel')}, strict=True)
    G = G.to(dev)
    B = 2
    lat = synth.make_latents(size, B, seed=6).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    img, _ = G(lat, input_is_latent=True, noise=noises)
    st = max(size // 128, 1)
    e = maxdiff(img[:, :, ::st, ::st], g['image'])
    e_m = max(maxdiff(img.double().mean(dim=(2, 3)), g['image_mean']), maxdiff(img.double().std(dim=(2, 3)), g['image_std']))
    padded = G._inner[0].engine().padded
    print(f'StyleGAN2Generator({size}, narrow={narrow}) vs reference: max |d| {e:.2e} on |image| <= {g["image"].abs().max().item():.2f}, moments {e_m:.2e}, zero-padded: {padded}')
    assert e < 1e-3 and e_m < 1e-5
    assert padded == any(int(c) % 16 for c in g['channels'])
    # return_features hands out the real channels only
    _, feat = G._inner[0](lat, input_is_tensor=True, input_is_latent=True, noise=noises, return_features=True)
    assert feat.shape[1] == int(g['channels'][-1])
    if not padded:
        return
    # the W+ gradient through the padded layers (exact step, then the carried-scale step with the fused producers) vs the oracle in float64
    P = ros
    Bg = 1
    w = synth.make_latents(size, Bg, seed=14).double().requires_grad_(True)
    nz = synth.make_noises(size, Bg, seed=7)
    target = synth.make_images(size, Bg, seed=9)
    img_ref = R.generator_forward({k: v.double() for k, v in P.items()}, w, [n.double() for n in nz], size)
    R.wplus_loss(img_ref, target.double()).backward()
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size, narrow=narrow)
    gmul = ops.loss_scale_for(3 * size * size)
    eng.reset_bwd_state()
    eng.reset_fwd_state()
    for rep in range(2):
        im = eng.forward(w.detach().float().to(dev), [n.to(dev) for n in nz], save=True, range_mode='carry')
        loss, gimg = ops.mse_loss_grad(im, target.to(dev), gmul)
        glat = eng.backward(gimg, gmul, carry_scale=True)
        rel = (glat.cpu().double() - w.grad).abs().max().item() / w.grad.abs().max().item()
        print(f'  rep{rep}: image {maxdiff(im, img_ref.detach().float()):.2e}, dL/dW+ rel {rel:.2e}')
        assert maxdiff(im, img_ref.detach().float()) < 1e-3 and rel < 3e-4
    assert not eng.bwd_scale_violated() and not eng.fwd_range_violated()
    with pytest.raises(NotImplementedError):
        eng.forward(w.detach().float().to(dev), [n.to(dev) for n in nz], cond_hook=lambda *a: None, cond_layers=[5])


@pytest.mark.parametrize('prec', ['f16s', 'f32'])
@pytest.mark.parametrize('size,B', [(16, 2), (64, 1)])
def test_generator_backward_vs_oracle_autograd(dev, size, B, prec):
    """dL/dW+ from the HIP backward kernels vs torch autograd through the oracle evaluated in
    float64.  The latent seed is chosen so that no pre-activation of the low-resolution layers lies
    within fp32 rounding of the LeakyReLU kink (seed 6 has one at 7e-8 in conv1: its sign — and with it
    0.4 % of dL/dw[0] — legitimately differs between fp32 implementations; scratch seed scan in DESIGN.md)."""
    from oodgan.engine import GeneratorEngine
    from oodgan import ops
    P = synth.generator_state(size, seed=5)
    lat = synth.make_latents(size, B, seed=14)
    noises = synth.make_noises(size, B, seed=7)
    target = synth.make_images(size, B, seed=9)
    w = lat.double().requires_grad_(True)
    img_ref = R.generator_forward({k: v.double() for k, v in P.items()}, w, [n.double() for n in noises], size)
    R.wplus_loss(img_ref, target.double()).backward()
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size, precision=prec)
    img = eng.forward(lat.to(dev), [n.to(dev) for n in noises], save=True)
    assert maxdiff(img, img_ref.detach().float()) < 1e-3
    gmul = ops.loss_scale_for(3 * size * size)
    loss, gimg = ops.mse_loss_grad(img, target.to(dev), gmul)
    glat = eng.backward(gimg, gmul)
    gref = w.grad
    rel = (glat.detach().cpu().double() - gref).abs().max().item() / gref.abs().max().item()
    assert rel < 1e-4, rel
    # range control: a 2^24 x larger (or smaller) loss scale must give the same gradient (the split-f16 convs
    # rescale every gradient tensor by a power of two derived from its max)
    for k in (2.0 ** 24, 2.0 ** -20):
        loss2, gimg2 = ops.mse_loss_grad(img, target.to(dev), gmul * k)
        glat2 = eng.backward(gimg2, gmul * k)
        assert torch.isfinite(glat2).all()
        rel2 = (glat2 - glat).abs().max().item() / gref.abs().max().item()
        assert rel2 < 1e-4, (k, rel2)


def test_wplus_trajectory_vs_golden(dev, golden):
    from oodgan.engine import GeneratorEngine, WPlusInverter
    g = golden('wplus_s32.npz')
    P = synth.generator_state(32, seed=5)
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, 32)
    target = synth.make_images(32, 2, seed=9).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(32, 2, seed=7)]
    w0 = synth.make_latents(32, 2, seed=6).to(dev)
    w, losses, traj = WPlusInverter(eng).invert(target, w0, noises, steps=5, return_trajectory=True)
    assert maxdiff(losses, g['losses']) < 1e-3 * g['losses'].abs().max().item()
    # Adam's first steps move every coordinate by ~lr regardless of gradient scale, so sign flips
    # of near-zero gradients are the only way to differ; compare the trajectory itself
    dw = (torch.stack(traj).cpu() - g['traj']).abs()
    assert (dw < 2e-3).float().mean().item() > 0.999, dw.max().item()
    assert losses[-1].sum() < losses[0].sum()


def test_wplus_streams_and_graph_match_single_stream(dev):
    """hipGraph replay of the W+ step must reproduce the eager single-stream loop; sub-batches advanced on separate
    HIP streams must be run-to-run bit-reproducible and agree with the single-stream loop to rounding (the carried
    range scales are per sub-batch, so the split-f16 operands may round differently in the last bit).

    History: with packed-fp32 VALU instructions in the kernels, two generator passes running
    concurrently on different HIP streams gave run-to-run deviations of up to 1.5e-3; the library is built without
    them and the deviation is gone — this test is the guard."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, B = 32, 4
    P = synth.generator_state(size, seed=5)
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
    target = synth.make_images(size, B, seed=9).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    w0 = synth.make_latents(size, B, seed=14).to(dev)
    w1, l1 = WPlusInverter(eng).invert(target, w0, noises, steps=6)
    # one stream, captured graph: identical arithmetic, identical order
    w2, l2 = WPlusInverter(eng).invert(target, w0, noises, steps=6, streams=1, use_graph=True)
    torch.cuda.synchronize()
    assert maxdiff(l2, l1.cpu()) <= 1e-5 * l1.abs().max().item()
    assert ((w2 - w1).abs() < 1e-4).float().mean().item() > 0.999
    for streams, graph in ((2, False), (3, False), (4, True)):
        runs = []
        for _ in range(3):
            w2, l2 = WPlusInverter(eng).invert(target, w0, noises, steps=6, streams=streams, use_graph=graph)
            torch.cuda.synchronize()
            runs.append((w2.clone(), l2.clone()))
        for w3, l3 in runs[1:]:
            assert torch.equal(w3, runs[0][0]) and torch.equal(l3, runs[0][1]), (streams, graph)
        w2, l2 = runs[0]
        assert l2.shape == l1.shape and torch.isfinite(w2).all()
        assert maxdiff(l2, l1.cpu()) <= 2e-5 * l1.abs().max().item(), (streams, graph)
        assert (l2[-1] < l2[0]).all()


@pytest.mark.gpu
def test_concurrent_streams_bit_reproducible_at_256(dev):
    """The production kernel instances (big-tile, strip-free 256² geometry) on 4 concurrent streams, five repetitions."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, B = 256, 8
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=5).items()}, size)
    target = synth.make_images(size, B, seed=1).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=2)]
    w0 = synth.make_latents(size, B, seed=3, std=0.5).to(dev)
    ref = None
    for _ in range(5):
        w, l = WPlusInverter(eng).invert(target, w0, noises, steps=4, streams=4)
        torch.cuda.synchronize()
        if ref is None:
            ref = (w.clone(), l.clone())
        assert torch.equal(w, ref[0]) and torch.equal(l, ref[1])
    w1, l1 = WPlusInverter(eng).invert(target, w0, noises, steps=4)
    assert maxdiff(ref[1], l1.cpu()) <= 2e-5 * l1.abs().max().item()


def test_full_size_inversion_properties(dev):
    """BASELINE configs[2] geometry (1024², every production kernel instance incl. the strip kernel of the 32-channel
    layers and the fused producers) through size-independent properties: (a) images are independent — inverting a
    sub-batch alone reproduces its part of the batch run; (b) the carried-scale fused backward follows the exact
    per-step scales; (c) the loss decreases."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, B, steps = 1024, 4, 3
    P = synth.generator_state(size, seed=0)
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
    target = torch.cat([synth.make_images(size, 1, seed=1000 + g) for g in range(B)]).to(dev)
    noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + g)[i] for g in range(B)]).to(dev) for i in range(17)]
    w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + g, std=0.3) for g in range(B)]).to(dev)
    inv = WPlusInverter(eng)
    w, l = inv.invert(target, w0, noises, steps=steps)
    assert torch.isfinite(w).all() and (l[-1] < l[0]).all() and not eng.bwd_scale_violated()
    ws, ls = inv.invert(target[:2].contiguous(), w0[:2].contiguous(), [n[:2].contiguous() for n in noises], steps=steps)
    assert maxdiff(ls, l[:, :2].cpu()) <= 1e-4 * l.abs().max().item()
    assert ((ws - w[:2]).abs() < 5e-4).float().mean().item() > 0.999
    eng.fused_bwd = False
    w2, l2 = inv.invert(target, w0, noises, steps=steps)
    eng.fused_bwd = True
    assert maxdiff(l2, l.cpu()) <= 2e-5 * l.abs().max().item()
    assert ((w2 - w).abs() < 1e-4).float().mean().item() > 0.999


@pytest.mark.parametrize('streams', [2, 3])
def test_streams_vs_one_at_full_size(dev, streams):
    """The bench configuration (1024², B=8; sub-batches of 4 / 4 images on two HIP streams — bench.py's default since round 4 —
    and of 2 / 3 / 3 on three, the round-3 default) against the single-stream
    loop: every production kernel — the 8-wave stride-1 / stride-2 (fused activation backward) / transposed kernels as
    neighbours, every HBM-bound producer (rgb_finish, blur_act_sform, act_bwd_*, ToRGB) as bystander — runs beside the other
    queues' work here.  (a) bit-identical run to run; (b) equal to one stream up to the fp32 rounding that the different
    kernel selection of a 2-3 image sub-batch and its own range scales bring."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, B, steps = 1024, 8, 3
    P = synth.generator_state(size, seed=0)
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
    target = torch.cat([synth.make_images(size, 1, seed=1000 + g) for g in range(B)]).to(dev)
    noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + g)[i] for g in range(B)]).to(dev) for i in range(17)]
    w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + g, std=0.3) for g in range(B)]).to(dev)
    inv = WPlusInverter(eng)
    w1, l1 = inv.invert(target, w0, noises, steps=steps, streams=1)
    runs = []
    for _ in range(3):
        w3, l3 = inv.invert(target, w0, noises, steps=steps, streams=streams)
        torch.cuda.synchronize()
        runs.append((w3.clone(), l3.clone()))
    assert all(torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) for r in runs[1:])
    w3, l3 = runs[0]
    # step 1 sees the same w: its loss differs by rounding only.  Adam's first steps move every coordinate by ~lr*sign(g), so a
    # handful of ~0 gradient coordinates take the other sign and the later losses follow them (measured 2e-4 relative)
    rel0 = (l3[0] - l1[0]).abs().max().item() / l1[0].abs().max().item()
    rel = (l3 - l1).abs().max().item() / l1.abs().max().item()
    frac = ((w3 - w1).abs() < 5e-4).float().mean().item()
    print(f'{streams} streams vs 1 at 1024², B=8: loss rel diff step 1 {rel0:.2e}, all steps {rel:.2e}, {100 * frac:.3f}% of w within 5e-4')
    assert rel0 <= 1e-5 and rel <= 1e-3 and frac > 0.999 and (l3[-1] < l3[0]).all()


def test_graph_replay_at_full_size(dev):
    """A W+ step captured into a hipGraph and replayed (WPlusInverter.invert(use_graph=True)) at the bench geometry: the F-form path
    of the 1024² level (conv_f16s_stripx.hip) is first used in step 2, i.e. inside the capture — its one-time allocations and
    attribute calls must have happened before (done when the engine is built).  Same trajectory as the eager loop."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, B, steps = 1024, 2, 4
    P = synth.generator_state(size, seed=0)
    eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
    target = torch.cat([synth.make_images(size, 1, seed=1000 + g) for g in range(B)]).to(dev)
    noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + g)[i] for g in range(B)]).to(dev) for i in range(17)]
    w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + g, std=0.3) for g in range(B)]).to(dev)
    inv = WPlusInverter(eng)
    w1, l1 = inv.invert(target, w0, noises, steps=steps, streams=1)
    w2, l2 = inv.invert(target, w0, noises, steps=steps, streams=1, use_graph=True)
    assert torch.equal(l1, l2) and torch.equal(w1, w2)


def test_torgb_reproducible_beside_matrix_kernels_of_another_stream(dev):
    """Guard for DESIGN.md §10: a ToRGB launch that shares the GPU with the stride-2 / transposed matrix kernels of
    another HIP stream must give the same bits as alone (with packed-fp32 instructions in the kernels 70–85 % of such
    launches came out with a few wrong elements)."""
    import math
    from oodgan import ops
    from oracle import ref_cpu as R
    g = torch.Generator().manual_seed(0)
    B = 4
    k = (R.make_kernel([1, 3, 3, 1]) * 4.0).to(dev)

    def conv_case(C, M, H, mode):
        x = torch.randn(B, C, H, H, generator=g).to(dev)
        s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
        d = (1 + 0.3 * torch.randn(B, M, generator=g)).to(dev)
        w = (torch.randn(M, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
        wpk = ops.pack_conv3x3(w, precision='f16s')
        if mode == ops.CONV_S2:
            Hin = 2 * H + 1
            P2 = (Hin + 3) // 4 * 4
            g2 = torch.randn(B, C, Hin, P2, generator=g).to(dev)
            xs = ops.to_sform_phases(g2, H, H, s, in_pitch=P2)
        else:
            xs = ops.to_sform(x, s)
        return lambda: ops.conv3x3(xs, wpk, M, mode, out_scale=d)

    neighbours = [conv_case(512, 512, 8, ops.CONV_S2), conv_case(512, 256, 64, ops.CONV_T2), conv_case(32, 32, 512, ops.CONV_S1)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for C0, H0 in ((512, 16), (64, 128)):                 # torgb_fwd_small_kernel / torgb_fwd_kernel
        x0 = torch.randn(B, C0, H0, H0, generator=g).to(dev)
        s0 = (1 + 0.3 * torch.randn(B, C0, generator=g)).to(dev)
        wr = torch.randn(3, C0, generator=g).to(dev)
        bias = torch.randn(3, generator=g).to(dev)
        skip = torch.randn(B, 3, H0 // 2, H0 // 2, generator=g).to(dev)
        torch.cuda.synchronize()
        ref = ops.torgb(x0, wr, s0, bias, skip, k).clone()
        torch.cuda.synchronize()
        for fn in neighbours:
            for _ in range(3):
                with torch.cuda.stream(sb):
                    for _ in range(12):
                        fn()
                with torch.cuda.stream(sa):
                    outs = [ops.torgb(x0, wr, s0, bias, skip, k) for _ in range(40)]
                torch.cuda.synchronize()
                assert all(torch.equal(o, ref) for o in outs)


def test_features_in_injection_vs_golden(dev, golden):
    """`insert_feature` (model.py:541-546): vectors produced by the reference Generator with features_in / feature_scale."""
    from oodgan.modules import Generator
    size, B = 32, 2
    g = golden('features_in_s32.npz')
    G = Generator(size, 512, 8)
    G.load_state_dict(synth.generator_state(size, seed=5), strict=True)
    G = G.to(dev).eval()
    lat = synth.make_latents(size, B, seed=6).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    feats = [None] * 8
    feats[4] = synth.normal('featin.4', (B, 512, 16, 16), 9).to(dev)
    feats[5] = synth.normal('featin.5', (B, 512, 16, 16), 10).to(dev)
    for fs in (0.3, 1.0):
        img, feat = G(lat, input_is_tensor=True, input_is_latent=True, noise=noises, return_features=True, features_in=feats,
                      feature_scale=fs)
        ref = g[f'image_fs{fs}']
        assert maxdiff(img, ref) <= 1e-4 * max(1.0, ref.abs().max().item()), fs
        assert maxdiff(feat[:, ::16], g[f'feat_fs{fs}_sub']) <= 1e-4 * max(1.0, g[f'feat_fs{fs}_sub'].abs().max().item())


def test_batched_backward_tail_is_bit_identical(dev):
    """The batched tail of the backward (partial-sum reductions, demodulation gradient, dot products, scale checks in four
    launches) must reproduce the per-layer launches bit for bit: same per-output summation order, same order of the two
    contributions to every accumulator entry (the demodulation kernel's += is a fused multiply-add)."""
    from oodgan.engine import GeneratorEngine, WPlusInverter
    size, B = 64, 3
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=5).items()}, size)
    target = synth.make_images(size, B, seed=9).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    w0 = synth.make_latents(size, B, seed=14).to(dev)
    res = []
    for flag in (False, True):
        eng.batched_tail = flag
        w, l = WPlusInverter(eng).invert(target, w0, noises, steps=5)
        torch.cuda.synchronize()
        res.append((w.clone(), l.clone()))
    eng.batched_tail = True
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_engine_cache_follows_weight_changes(dev):
    """The prepared-weights engine must be rebuilt when the generator's weights change by ANY route: load_state_dict on a
    PARENT module (recurses through _load_from_state_dict, never calls Generator.load_state_dict — what BasicSR's
    load_network / resume does) and in-place writes."""
    from oodgan.modules import Generator

    class Wrapper(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.generator = Generator(32, 512, 8)

    m = Wrapper()
    m.load_state_dict({'generator.' + k: v for k, v in synth.generator_state(32, seed=5).items()}, strict=True)
    m = m.to(dev).eval()
    lat = synth.make_latents(32, 2, seed=6).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(32, 2, seed=7)]
    run = lambda: m.generator(lat, input_is_tensor=True, input_is_latent=True, noise=noises)[0].clone()
    img_a = run()
    eng_a = m.generator.engine()
    assert m.generator.engine() is eng_a                      # unchanged weights: cached
    m.load_state_dict({'generator.' + k: v for k, v in synth.generator_state(32, seed=6).items()}, strict=True)
    img_b = run()
    fresh = Generator(32, 512, 8)
    fresh.load_state_dict(synth.generator_state(32, seed=6), strict=True)
    fresh = fresh.to(dev).eval()
    ref_b = fresh(lat, input_is_tensor=True, input_is_latent=True, noise=noises)[0]
    assert (img_b - img_a).abs().max().item() > 1e-2 and torch.equal(img_b, ref_b)
    with torch.no_grad():
        m.generator.conv1.conv.weight.mul_(0.5)               # in-place write
        m.generator.conv1.activate.bias.add_(0.25)
    img_c = run()
    assert (img_c - img_b).abs().max().item() > 1e-3


def test_cond_types_sft_add_fuse_vs_golden(dev, golden):
    """Generator.forward / StyleGAN2Generator.forward with ``conditions`` for cond_type 'SFT', 'ADD' (with and without the
    callback) and 'FUSE' (feature_modulation, model.py:558-566,588-610; stylegan2_arch.py:583-595) against vectors
    produced by the reference Generator."""
    from oodgan.modules import Generator, StyleGAN2Generator
    size, B = 32, 2
    g = golden('cond_types_s32.npz')
    ros = synth.generator_state(size, seed=5)
    G = Generator(size, 512, 8)
    G.load_state_dict(ros, strict=True)
    G = G.to(dev).eval()
    G2 = StyleGAN2Generator(size)
    G2.load_state_dict({G2._ros_to_basicsr(k): v for k, v in ros.items() if not k.endswith('.kernel')}, strict=True)
    G2 = G2.to(dev)
    lat = synth.make_latents(size, B, seed=6).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    conds = lambda: [[synth.normal(f'cond.{k}.0', (B, 512, r, r), 9, 0.5).to(dev), synth.normal(f'cond.{k}.1', (B, 512, r, r), 10, 0.5).to(dev)]
                     for k, r in ((0, 8), (1, 16))]
    for ct in ('SFT', 'ADD', 'FUSE'):
        img, feat = G(lat, input_is_tensor=True, input_is_latent=True, noise=noises, return_features=True, conditions=conds(),
                      cond_layers=[1, 3], cond_type=ct)
        ref = g[f'image_{ct}']
        assert maxdiff(img, ref) <= 1e-4 * max(1.0, ref.abs().max().item()), ct
        assert maxdiff(feat[:, ::16], g[f'feat_{ct}_sub']) <= 1e-4 * max(1.0, g[f'feat_{ct}_sub'].abs().max().item())
        img2, _ = G2(lat, input_is_latent=True, noise=noises, conditions=conds(), cond_layers=[1, 3], cond_type=ct)
        assert maxdiff(img2, ref) <= 1e-4 * max(1.0, ref.abs().max().item()), ct
    cb = lambda feats, **kw: conds()[kw['index']][0] * 0.25 + kw['style'][:, :1, None, None]
    img, _ = G(lat, input_is_tensor=True, input_is_latent=True, noise=noises, conditions=conds(), cond_layers=[1, 3], cond_type='ADD', callback=cb)
    assert maxdiff(img, g['image_ADD_callback']) <= 1e-4 * max(1.0, g['image_ADD_callback'].abs().max().item())
    with pytest.raises(NotImplementedError):
        G2(lat, input_is_latent=True, noise=noises, conditions=conds(), cond_layers=[1, 3], cond_type='NOISE')


@pytest.mark.parametrize('prec', ['f16s', 'f32'])
def test_generator_1024_batch4_vs_reference(dev, golden, prec, monkeypatch):
    """BASELINE configs[1] (C2) literally: StyleGAN2 1024² generator forward, batch 4 — ``Generator([z], noise=<list>)``, mapping
    MLP included — against the reference Generator's own fp32 output on the bench recipe's weights (tests/golden/make_golden.py
    gold_generator_1024_b4; the reference's fp32 is 2e-5 from its own float64).  Both arithmetic variants of the conv kernels."""
    from oodgan import ops
    from oodgan.modules import Generator
    from make_golden_params import GEN_B4
    g = golden('generator_1024_b4.npz')
    size, B = GEN_B4['size'], GEN_B4['batch']
    monkeypatch.setattr(ops, 'PRECISION', prec)
    G = Generator(size, 512, 8)
    G.load_state_dict(synth.generator_state(size, seed=0), strict=True)
    G = G.to(dev).eval()
    z = synth.normal('gen_b4.z', (B, 512), GEN_B4['z_seed']).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=GEN_B4['noise_seed'])]
    img, lat = G([z], noise=noises, return_latents=True)
    assert G.engine().precision == prec and img.shape == (B, 3, size, size)
    assert maxdiff(lat[:, 0], g['latent']) < 1e-4 and torch.equal(lat[:, 0], lat[:, 17])
    im = img.double().cpu()
    e_sub = (im[:, :, ::16, ::16] - g['image_sub']).abs().max().item()
    e_crop = (im[:, :, 448:512, 512:576] - g['image_crop']).abs().max().item()
    e_mom = max((im.mean(dim=(2, 3)) - g['image_mean']).abs().max().item(), (im.std(dim=(2, 3)) - g['image_std']).abs().max().item())
    print(f'[{prec}] C2 generator forward 1024² B=4 vs reference fp32: sub-sample {e_sub:.2e}, crop {e_crop:.2e}, moments {e_mom:.2e} '
          f'(absmax {g["image_absmax"].item():.2f}; reference fp32 vs f64 {g["ref_f32_vs_f64"].item():.1e})')
    assert e_sub < 1e-3 and e_crop < 1e-3 and e_mom < 1e-5


def test_generator_forward_carried_scales_and_fallback(dev):
    """``Generator.forward`` with modules.CARRY_FORWARD (opt-in since round 6) carries the forward range scales from call to call.  (0) by DEFAULT the
    forward is a pure function of its inputs: a call after other inputs equals a fresh generator's call bit for bit.  (a) with the switch on the second call
    (carried scales, fused producers) equals the first (measured scales) to fp32 rounding and the third equals the second bit for bit;
    (b) an input whose activations leave the carried window — noise maps 2^30 times larger — sets the check flag and the pass is repeated
    with measured scales: same result as a generator that never carried anything."""
    from oodgan import modules
    from oodgan.modules import Generator
    size, B = 256, 2
    sd = synth.generator_state(size, seed=3)

    def build():
        G = Generator(size, 512, 8)
        G.load_state_dict(sd, strict=True)
        return G.to(dev).eval()
    z = synth.normal('carry.z', (B, 512), 5).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=6)]
    big = [n * 2.0 ** 30 for n in noises]
    assert modules.CARRY_FORWARD is False                # the shipped default
    G0 = build()
    cold, _ = G0([z], noise=noises)
    cold = cold.clone()
    G0([z * 3.0], noise=big)                             # something else in between ...
    again, _ = G0([z], noise=noises)
    fresh, _ = build()([z], noise=noises)
    assert torch.equal(again, cold) and torch.equal(fresh, cold)      # ... leaves no trace
    old = modules.CARRY_FORWARD
    try:
        modules.CARRY_FORWARD = True
        G = build()
        img1, _ = G([z], noise=noises)
        img1 = img1.clone()
        assert torch.equal(img1, cold)                   # the first call of a carrying generator measures: same as the default
        eng = G.engine()
        assert eng.fwd_range is not None and eng.fwd_range.valid
        img2, _ = G([z], noise=noises)
        img2 = img2.clone()
        img3, _ = G([z], noise=noises)
        assert maxdiff(img2, img1.cpu()) <= 2e-5 * img1.abs().max().item() and torch.equal(img3, img2)
        out_big, _ = G([z], noise=big)                       # carried scales are 2^30 off: violation -> measured-scale repeat
        assert not eng.fwd_range_violated() and torch.isfinite(out_big).all()
        modules.CARRY_FORWARD = False
        ref_big, _ = build()([z], noise=big)
        modules.CARRY_FORWARD = True
        assert maxdiff(out_big, ref_big.cpu()) <= 2e-5 * ref_big.abs().max().item()
        back, _ = G([z], noise=noises)                       # and back again (2^-30): violation the other way, or a valid carried window
        assert maxdiff(back, img1.cpu()) <= 2e-5 * img1.abs().max().item()
    finally:
        modules.CARRY_FORWARD = old


@pytest.mark.gpu
def test_backward_refuses_saved_activations_that_a_later_forward_may_have_overwritten(dev):
    """the saved activations live in pooled scratch buffers; a forward between forward(save=True) and backward() may overwrite
    them — the engine refuses instead of computing a gradient from the wrong tensors."""
    from oodgan.engine import GeneratorEngine
    from oodgan import ops
    size, B = 32, 2
    eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=5).items()}, size)
    lat = synth.make_latents(size, B, seed=14).to(dev)
    noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
    img = eng.forward(lat, noises, save=True)
    gmul = ops.loss_scale_for(3 * size * size)
    _, gimg = ops.mse_loss_grad(img, torch.zeros_like(img), gmul)
    g0 = eng.backward(gimg, gmul)                        # the normal order works
    assert torch.isfinite(g0).all()
    eng.forward(lat, noises, save=True)
    eng.forward(lat, noises)                             # e.g. a model(x) call in between
    with pytest.raises(RuntimeError, match='another forward'):
        eng.backward(gimg, gmul)
