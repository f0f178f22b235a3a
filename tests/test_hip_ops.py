"""GPU parity: every HIP kernel (through the C ABI) against the CPU oracle on the same seeded inputs
and against the committed golden vectors.  Tolerance: |diff| <= 1e-3 absolute in fp32 (north star),
in practice ~1e-5; integer/index work is exact."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oodgan import synth  # noqa: E402

TOL = 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a ROCm device'
    return torch.device('cuda:0')


def close(a, b, tol=TOL):
    a = a.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = max(1.0, b.abs().max().item())
    assert err <= tol * ref, f'max abs err {err:.3e} (ref max {ref:.3e})'


def test_upfirdn2d_golden_and_oracle(dev, golden):
    from oodgan import ops
    g = golden('ops.npz')
    k4 = R.make_kernel([1, 3, 3, 1])
    x = g['ufd_x']
    for tag, kern, up, down, pad in [
        ('blur11', k4 * 4, 1, 1, (1, 1)), ('up2', k4 * 4, 2, 1, (2, 1)), ('blur21', k4, 1, 1, (2, 1)),
        ('down2', k4 * 4, 1, 2, (1, 2)), ('blur22', k4 * 4, 1, 1, (2, 2)), ('crop', k4, 1, 1, (-1, 3)),
    ]:
        close(ops.upfirdn2d(x.to(dev), kern.to(dev), up, down, pad), g[f'ufd_{tag}'])
    # tiled path (>= 64x64 outputs), ragged sizes, asymmetric kernel (checks the flip)
    ka = synth.normal('t.k', (4, 4), 1)
    for shape, pad in [((2, 3, 70, 131), (2, 1)), ((1, 2, 129, 65), (1, 1)), ((1, 1, 64, 64), (2, 2)), ((1, 2, 97, 200), (0, 3))]:
        xx = synth.normal('t.x', shape, 2)
        close(ops.upfirdn2d(xx.to(dev), ka.to(dev), 1, 1, pad), R.upfirdn2d(xx, ka, 1, 1, pad))
    k3 = synth.normal('t.k3', (3, 2), 1)
    xx = synth.normal('t.x2', (2, 2, 33, 47), 3)
    for up, down, pad in [(1, 1, (1, 1)), (2, 1, (1, 2)), (1, 2, (2, 0)), (3, 2, (2, 2))]:
        close(ops.upfirdn2d(xx.to(dev), k3.to(dev), up, down, pad), R.upfirdn2d(xx, k3, up, down, pad))


def test_upfirdn2d_empty_batch(dev):
    from oodgan import ops
    y = ops.upfirdn2d(torch.zeros(0, 3, 8, 8, device=dev), torch.ones(4, 4, device=dev), pad=(2, 1))
    assert y.shape == (0, 3, 8, 8)


def test_fused_leaky_relu(dev, golden):
    from oodgan import ops
    g = golden('ops.npz')
    close(ops.fused_leaky_relu(g['ufd_x'].to(dev), g['flr_b'].to(dev)), g['flr_y'])
    close(ops.fused_leaky_relu(g['ufd_x'].to(dev), g['flr_b'].to(dev), 0.1, 1.5), g['flr_y2'])
    # 2-D input (mapping network) and backward (case 31 of the reference kernel)
    x = synth.normal('f.x', (5, 24), 1)
    b = synth.normal('f.b', (24,), 1)
    close(ops.fused_leaky_relu(x.to(dev), b.to(dev)), R.fused_leaky_relu(x, b))
    x4 = synth.normal('f.x4', (2, 6, 9, 13), 1).requires_grad_(True)
    b4 = synth.normal('f.b4', (6,), 1).requires_grad_(True)
    y = R.fused_leaky_relu(x4, b4)
    gy = synth.normal('f.gy', tuple(y.shape), 1)
    y.backward(gy)
    gx, gb = ops.fused_leaky_relu_backward(gy.to(dev), y.detach().to(dev), need_bias_grad=True)
    close(gx, x4.grad)
    close(gb, b4.grad)


def test_equal_linear_and_mapping(dev, golden):
    from oodgan import ops
    g = golden('ops.npz')
    close(ops.equal_linear(g['lin_x'].to(dev), g['lin_w'].to(dev), g['lin_b'].to(dev), 0.01, False), g['lin_y'])
    close(ops.equal_linear(g['lin_x'].to(dev), g['lin_w'].to(dev), g['lin_b'].to(dev), 0.01, True), g['lin_y_act'])
    z = synth.normal('pn', (3, 512), 1)
    close(ops.pixel_norm(z.to(dev)), z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8))


def test_style_affine_mfma_and_generic(dev):
    from oodgan import ops
    B, L, S = 5, 3, 64
    lat = synth.normal('sa.lat', (B, L, S), 1)
    for R_, rl in [(48, [0] * 16 + [2] * 32), (20, [1] * 7 + [0] * 13)]:
        w = synth.normal('sa.w', (R_, S), 2)
        b = synth.normal('sa.b', (R_,), 2)
        row_lat = torch.tensor(rl, dtype=torch.int32)
        ref = torch.stack([R.equal_linear(lat[:, rl[r]], w[r:r + 1], b[r:r + 1])[:, 0] for r in range(R_)], dim=1)
        close(ops.style_affine(lat.to(dev), w.to(dev), b.to(dev), row_lat.to(dev)), ref)
    # backward: rows grouped by latent
    R_, rl = 48, [0] * 16 + [2] * 32
    w = synth.normal('sa.w', (R_, S), 2)
    gs = synth.normal('sa.gs', (B, R_), 3)
    lat_start = torch.tensor([0, 16, 16, 48], dtype=torch.int32)
    ref = torch.zeros(B, L, S)
    for r in range(R_):
        ref[:, rl[r]] += gs[:, r:r + 1] * w[r] / math.sqrt(S)
    close(ops.style_affine_backward(gs.to(dev), w.to(dev), lat_start.to(dev), L), ref)


def _mc_ref(g, tag, demod, ups):
    return R.modulated_conv2d(g['mc_x'], g['mc_wlat'], g[f'mc_{tag}_w'], g[f'mc_{tag}_mw'], g[f'mc_{tag}_mb'], demod, ups)


def test_modulated_conv_modules_vs_golden(dev, golden):
    from oodgan.modules import ModulatedConv2d, StyledConv, ToRGB
    g = golden('ops.npz')
    B, Ci, Co, H, S = 2, 16, 8, 12, 64
    x, wl = g['mc_x'].to(dev), g['mc_wlat'].to(dev)
    for tag, k, demod, ups in [('plain', 3, True, False), ('up', 3, True, True), ('rgb', 1, False, False)]:
        cout = 3 if tag == 'rgb' else Co
        mc = ModulatedConv2d(Ci, cout, k, S, demodulate=demod, upsample=ups)
        mc.weight.data = g[f'mc_{tag}_w']
        mc.modulation.weight.data = g[f'mc_{tag}_mw']
        mc.modulation.bias.data = g[f'mc_{tag}_mb']
        mc = mc.to(dev)
        close(mc(x, wl), g[f'mc_{tag}_y'])
    for tag, ups in [('sc', False), ('scup', True)]:
        sc = StyledConv(Ci, Co, 3, S, upsample=ups)
        sd = {}
        synth._styled_conv(sd, 'q', Ci, Co, S, 13, ups, 0.1)
        sc.load_state_dict({k_[2:]: v for k_, v in sd.items()})
        sc = sc.to(dev)
        close(sc(x, wl, noise=g[f'{tag}_noise'].to(dev)), g[f'{tag}_y'])
    rgb = ToRGB(Ci, S)
    sd = {}
    synth._to_rgb(sd, 'q', Ci, S, 13, True)
    rgb.load_state_dict({k_[2:]: v for k_, v in sd.items()})
    rgb = rgb.to(dev)
    close(rgb(x, wl, g['rgb_skip'].to(dev)), g['rgb_y'])
    close(rgb(x, wl, None), g['rgb_y_noskip'])


@pytest.mark.parametrize('prec', ['f32', 'f16s'])
@pytest.mark.parametrize('B,Ci,Co,H,W', [(1, 8, 32, 4, 4), (2, 24, 40, 9, 37), (1, 64, 64, 16, 16), (2, 32, 96, 33, 65),
                                          (1, 512, 64, 8, 8)])
def test_conv3x3_modes_vs_oracle(dev, B, Ci, Co, H, W, prec):
    """Raw implicit-GEMM kernels (ragged sizes, channel counts that are not multiples of the tile)
    against F.conv2d / conv_transpose2d with in/out scales and the dot epilogue, for both arithmetic
    variants: exact fp32 MFMA and split-f16 (3 MFMAs per product, fp32-equivalent)."""
    import torch.nn.functional as F
    from oodgan import ops
    x = synth.normal('cv.x', (B, Ci, H, W), 1)
    w = synth.normal('cv.w', (Co, Ci, 3, 3), 2, 1.0 / math.sqrt(Ci * 9))
    s = synth.normal('cv.s', (B, Ci), 3, 0.3, 1.0)
    d = synth.normal('cv.d', (B, Co), 4, 0.3, 1.0)
    xs = x * s[:, :, None, None]
    # S1 forward with scales
    ref = F.conv2d(xs, w, padding=1) * d[:, :, None, None]
    wpk = ops.pack_conv3x3(w.to(dev), precision=prec)
    close(ops.conv3x3(x.to(dev), wpk, Co, ops.CONV_S1, in_scale=s.to(dev), out_scale=d.to(dev)), ref)
    # S1 with fused noise + bias + lrelu
    nz = synth.normal('cv.nz', (B, 1, H, W), 5)
    nw = torch.tensor([0.37])
    bias = synth.normal('cv.b', (Co,), 6)
    ref2 = R.fused_leaky_relu(ref + nw * nz, bias)
    close(ops.conv3x3(x.to(dev), wpk, Co, ops.CONV_S1, in_scale=s.to(dev), out_scale=d.to(dev), bias=bias.to(dev),
                      noise=nz.to(dev), noise_weight=nw.to(dev), act=ops.ACT_LRELU), ref2)
    # T2 (transposed, stride 2): pitched output (B,Co,2H+1,2W+2)
    reft = F.conv_transpose2d(xs, w.transpose(0, 1), stride=2) * d[:, :, None, None]
    z = ops.conv3x3(x.to(dev), wpk, Co, ops.CONV_T2, in_scale=s.to(dev), out_scale=d.to(dev))
    assert z.shape == (B, Co, 2 * H + 1, (2 * W + 1 + 3) // 4 * 4)
    close(z[..., :2 * W + 1], reft)
    # input gradient of S1 (= S1 with transposed+flipped weights) with the style-gradient dot epilogue
    gy = synth.normal('cv.gy', (B, Co, H, W), 7)
    xs_ = xs.clone().requires_grad_(True)
    (F.conv2d(xs_, w, padding=1) * d[:, :, None, None] * gy).sum().backward()
    wpk_b = ops.pack_conv3x3(w.to(dev), 1.0, transpose=True, flip=True, precision=prec)
    dx, dot = ops.conv3x3(gy.to(dev), wpk_b, Ci, ops.CONV_S1, in_scale=d.to(dev), out_scale=s.to(dev), dotx=x.to(dev))
    close(dx, xs_.grad * s[:, :, None, None], 2e-4)
    close(dot, (xs_.grad * x).sum(dim=(2, 3)), 2e-4)
    # S2 = input gradient of T2
    gz = synth.normal('cv.gz', (B, Co, 2 * H + 1, 2 * W + 1), 8)
    xs_ = xs.clone().requires_grad_(True)
    (F.conv_transpose2d(xs_, w.transpose(0, 1), stride=2) * d[:, :, None, None] * gz).sum().backward()
    wpk_t = ops.pack_conv3x3(w.to(dev), 1.0, transpose=True, flip=False, precision=prec)
    gzp = torch.zeros(B, Co, 2 * H + 1, 2 * W + 2)
    gzp[..., :2 * W + 1] = gz
    dx, dot = ops.conv3x3(gzp.to(dev), wpk_t, Ci, ops.CONV_S2, in_scale=d.to(dev), out_scale=s.to(dev), dotx=x.to(dev),
                          in_hw=(2 * H + 1, 2 * W + 1), in_pitch=2 * W + 2)
    close(dx, xs_.grad * s[:, :, None, None], 2e-4)
    close(dot, (xs_.grad * x).sum(dim=(2, 3)), 2e-4)


def test_torgb_and_blur_bias_act(dev):
    from oodgan import ops
    B, Ci, H = 2, 40, 18
    x = synth.normal('tr.x', (B, Ci, H, H), 1)
    w = synth.normal('tr.w', (3, Ci), 2)
    s = synth.normal('tr.s', (B, Ci), 3, 0.3, 1.0)
    bias = synth.normal('tr.b', (3,), 4)
    skip = synth.normal('tr.skip', (B, 3, H // 2, H // 2), 5)
    k = R.make_kernel([1, 3, 3, 1]) * 4
    ref = torch.einsum('kc,bc,bchw->bkhw', w, s, x) / math.sqrt(Ci) + bias.view(1, 3, 1, 1) + R.upfirdn2d(skip, k, up=2, pad=(2, 1))
    close(ops.torgb(x.to(dev), w.to(dev), s.to(dev), bias.to(dev), skip.to(dev), k.to(dev)), ref)
    z = synth.normal('bb.z', (B, 6, 67, 67), 6)
    nz = synth.normal('bb.nz', (B, 1, 66, 66), 7)
    nw = torch.tensor([0.2])
    b6 = synth.normal('bb.b', (6,), 8)
    ref = R.fused_leaky_relu(R.upfirdn2d(z, k, pad=(1, 1)) + nw * nz, b6)
    close(ops.blur_bias_act(z.to(dev), k.to(dev), (1, 1), b6.to(dev), nz.to(dev), nw.to(dev)), ref)


def test_mse_and_adam(dev):
    from oodgan import ops
    a = synth.normal('m.a', (3, 3, 20, 20), 1).requires_grad_(True)
    t = synth.normal('m.t', (3, 3, 20, 20), 2)
    per = ((a - t) ** 2).mean(dim=(1, 2, 3))
    per.sum().backward()
    loss, g = ops.mse_loss_grad(a.detach().to(dev), t.to(dev))
    close(loss, per.detach(), 1e-5)
    close(g, a.grad, 1e-5)
    w = synth.normal('ad.w', (2, 18, 512), 3).requires_grad_(True)
    opt = torch.optim.Adam([w], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    wd = w.detach().clone().to(dev)
    m, v = torch.zeros_like(wd), torch.zeros_like(wd)
    for step in range(1, 4):
        gr = synth.normal(f'ad.g{step}', (2, 18, 512), 4) * (10.0 ** (-step))
        w.grad = gr.clone()
        opt.step()
        ops.adam_step(wd, gr.to(dev), m, v, step)
        close(wd, w.detach(), 1e-6)


@pytest.mark.parametrize('B,Ci,Co,H,W', [(1, 16, 32, 4, 4), (2, 24, 40, 9, 37), (2, 64, 64, 16, 16), (1, 32, 96, 33, 64), (1, 40, 24, 10, 36)])
def test_conv3x3_sform_input_and_output(dev, B, Ci, Co, H, W):
    """S-form path: F->S conversion (style folded in), S1 conv by LDS-DMA, fused epilogue, dot epilogue, and the
    S-form output consumed by a second conv (the conv->conv hand-off of the generator)."""
    import torch.nn.functional as F
    from oodgan import ops
    x = synth.normal('sf.x', (B, Ci, H, W), 1)
    w = synth.normal('sf.w', (Co, Ci, 3, 3), 2, 1.0 / math.sqrt(Ci * 9))
    s = synth.normal('sf.s', (B, Ci), 3, 0.3, 1.0)
    d = synth.normal('sf.d', (B, Co), 4, 0.3, 1.0)
    nz = synth.normal('sf.nz', (B, 1, H, W), 5)
    nw = torch.tensor([0.37])
    bias = synth.normal('sf.b', (Co,), 6)
    xs = ops.to_sform(x.to(dev), s.to(dev))
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    ref = R.fused_leaky_relu(F.conv2d(x * s[:, :, None, None], w, padding=1) * d[:, :, None, None] + nw * nz, bias)
    s2 = synth.normal('sf.s2', (B, Co), 7, 0.3, 1.0)
    ys = ops.SForm(B, Co, H, W, dev) if Co % 16 == 0 else None
    y = ops.conv3x3(xs, wpk, Co, ops.CONV_S1, out_scale=d.to(dev), bias=bias.to(dev), noise=nz.to(dev), noise_weight=nw.to(dev),
                    act=ops.ACT_LRELU, ys=ys, ys_scale=None if ys is None else s2.to(dev))
    close(y, ref)
    dotx = synth.normal('sf.dx', (B, Co, H, W), 8)
    y2, dot = ops.conv3x3(xs, wpk, Co, ops.CONV_S1, out_scale=d.to(dev), dotx=dotx.to(dev))
    raw = F.conv2d(x * s[:, :, None, None], w, padding=1)
    close(y2, raw * d[:, :, None, None])
    close(dot, (raw * dotx).sum(dim=(2, 3)), 2e-4)
    # transposed stride-2 conv from the same S-form input
    reft = F.conv_transpose2d(x * s[:, :, None, None], w.transpose(0, 1), stride=2) * d[:, :, None, None]
    zt = ops.conv3x3(xs, wpk, Co, ops.CONV_T2, out_scale=d.to(dev))
    close(zt[..., :2 * W + 1], reft)
    # stride-2 conv (input gradient of T2) on the phase-split S-form, with the dot epilogue
    gz = synth.normal('sf.gz', (B, Co, 2 * H + 1, 2 * W + 1), 11)
    xs_ = (x * s[:, :, None, None]).clone().requires_grad_(True)
    (F.conv_transpose2d(xs_, w.transpose(0, 1), stride=2) * d[:, :, None, None] * gz).sum().backward()
    wpk_t = ops.pack_conv3x3(w.to(dev), 1.0, transpose=True, flip=False, precision='f16s')
    gp = ops.to_sform_phases(gz.to(dev), H, W, d.to(dev))
    dxs, dots = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s.to(dev), dotx=x.to(dev))
    close(dxs, xs_.grad * s[:, :, None, None], 2e-4)
    close(dots, (xs_.grad * x).sum(dim=(2, 3)), 2e-4)
    if H % 2 == 0 and W % 4 == 0:
        # fused blur^T + phase split producer == upfirdn2d(pad=(2,2)) followed by the plain conversion
        k4 = R.make_kernel([1, 3, 3, 1]) * 4
        gq = synth.normal('sf.gq', (B, Co, 2 * H, 2 * W), 12)
        g2r = R.upfirdn2d(gq, torch.flip(k4, [0, 1]), pad=(2, 2))
        gp2 = ops.blurT_to_sform_phases(gq.to(dev), torch.flip(k4, [0, 1]).contiguous().to(dev), d.to(dev))
        gp3 = ops.to_sform_phases(g2r.to(dev), H, W, d.to(dev))
        assert (gp2.data.float() - gp3.data.float()).abs().max().item() < 2e-3 * g2r.abs().max().item()
        dxa = ops.conv3x3(gp2, wpk_t, Ci, ops.CONV_S2)
        dxb = ops.conv3x3(gp3, wpk_t, Ci, ops.CONV_S2)
        close(dxa, dxb.cpu(), 1e-5)
    if ys is not None:
        w2 = synth.normal('sf.w2', (Co, Co, 3, 3), 9, 1.0 / math.sqrt(Co * 9))
        wpk2 = ops.pack_conv3x3(w2.to(dev), precision='f16s')
        z = ops.conv3x3(ys, wpk2, Co, ops.CONV_S1)
        close(z, F.conv2d(ref * s2[:, :, None, None], w2, padding=1), 2e-4)


@pytest.mark.parametrize('B,C,H,W,act', [(2, 32, 8, 8, True), (1, 40, 9, 20, True), (1, 16, 4, 36, False),
                                         (2, 32, 32, 32, True), (1, 40, 37, 50, True), (1, 16, 33, 47, False), (1, 16, 80, 32, True)])
def test_blur_act_sform_vs_two_pass_and_oracle(dev, B, C, H, W, act):
    """Fused up-conv tail: y equals the oracle's blur + noise + bias + lrelu, and the S-form output equals
    to_sform(y, next style).  The kernel also records the range maximum of the S-form values."""
    from oodgan import ops
    seed = 23 + C
    Hz, Wz = 2 * H + 1, 2 * W + 1
    pitch = (Wz + 3) // 4 * 4
    z = synth.normal('z', (B, C, Hz, Wz), seed)
    zp = torch.full((B, C, Hz, pitch), float('nan'))          # the pitch padding must never be read as data
    zp[..., :Wz] = z
    noise = synth.normal('nz', (B, 1, 2 * H, 2 * W), seed)
    bias = synth.normal('b', (C,), seed, 0.2)
    nw = torch.tensor([0.3])
    s_next = synth.normal('s', (B, C), seed, 0.3, 1.0)
    k = R.make_kernel([1, 3, 3, 1]) * 4.0
    ref = R.upfirdn2d(z, k, pad=(1, 1)) + nw * noise
    ref = R.fused_leaky_relu(ref, bias) if act else ref + bias.reshape(1, -1, 1, 1)
    ys = ops.SForm(B, C, 2 * H, 2 * W, dev)
    y = ops.blur_act_sform(zp.to(dev), k.to(dev), H, W, bias.to(dev), noise.to(dev), nw.to(dev), act=act, ys=ys,
                           ys_scale=s_next.to(dev))
    close(y, ref)
    ref_s = ops.to_sform(y, s_next.to(dev))
    assert torch.equal(ys.data, ref_s.data)
    vm = torch.zeros(B, ops.VMAX_SLOTS, dtype=torch.int32, device=dev)
    ops.blur_act_sform(zp.to(dev), k.to(dev), H, W, bias.to(dev), noise.to(dev), nw.to(dev), act=act, ys=ys,
                       ys_scale=s_next.to(dev), vmax=vm)
    want = (y * s_next.to(dev)[:, :, None, None]).abs().amax(dim=(1, 2, 3))
    assert torch.equal(vm.view(torch.float32).amax(dim=1), want)
    # non-separable kernel -> generic 16-tap path
    k2 = k.clone()
    k2[1, 2] += 0.37
    y2 = ops.blur_act_sform(zp.to(dev), k2.to(dev), H, W, None, None, None, act=False)
    close(y2, R.upfirdn2d(z, k2, pad=(1, 1)))


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 32, 32, 16, 32), (1, 32, 32, 40, 64), (2, 24, 20, 13, 45), (1, 32, 32, 7, 100), (1, 30, 32, 72, 33)])
def test_conv3x3_strip_kernel_low_channel_layers(dev, B, Ci, Co, H, W):
    """17..32 -> 17..32 channel stride-1 convs on an S-form input take the strip-walking kernel (conv_f16s_strip.hip):
    forward epilogue (demod, noise, bias, lrelu), plain output, and the input-gradient instance with the dot epilogue
    and a power-of-two input range scale."""
    import torch.nn.functional as F
    from oodgan import ops
    x = synth.normal('st.x', (B, Ci, H, W), 1)
    w = synth.normal('st.w', (Co, Ci, 3, 3), 2, 1.0 / math.sqrt(Ci * 9))
    s = synth.normal('st.s', (B, Ci), 3, 0.3, 1.0)
    d = synth.normal('st.d', (B, Co), 4, 0.3, 1.0)
    nz = synth.normal('st.nz', (B, 1, H, W), 5)
    nw = torch.tensor([0.37])
    bias = synth.normal('st.b', (Co,), 6)
    xs = ops.to_sform(x.to(dev), s.to(dev))
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    raw = F.conv2d(x * s[:, :, None, None], w, padding=1)
    ref = R.fused_leaky_relu(raw * d[:, :, None, None] + nw * nz, bias)
    y = ops.conv3x3(xs, wpk, Co, ops.CONV_S1, out_scale=d.to(dev), bias=bias.to(dev), noise=nz.to(dev), noise_weight=nw.to(dev),
                    act=ops.ACT_LRELU)
    close(y, ref)
    close(ops.conv3x3(xs, wpk, Co, ops.CONV_S1, out_scale=d.to(dev)), raw * d[:, :, None, None])
    # shared noise (batch 1)
    y1 = ops.conv3x3(xs, wpk, Co, ops.CONV_S1, noise=nz[:1].to(dev), noise_weight=nw.to(dev))
    close(y1, raw + nw * nz[:1])
    # backward instance: input pre-scaled by 2^7 (mul2 = {2^-7, 2^7}), dot with the saved forward input
    mul2 = torch.tensor([2.0 ** -7, 2.0 ** 7], device=dev)
    xs2 = ops.to_sform(x.to(dev), s.to(dev), mul2)
    dotx = synth.normal('st.dx', (B, Co, H, W), 8)
    y2, dot = ops.conv3x3(xs2, wpk, Co, ops.CONV_S1, out_scale=d.to(dev), dotx=dotx.to(dev), in_mul2=mul2)
    close(y2, raw * d[:, :, None, None])
    close(dot, (raw * dotx).sum(dim=(2, 3)), 2e-4)
    # run-to-run identical (deterministic reductions)
    y3, dot3 = ops.conv3x3(xs2, wpk, Co, ops.CONV_S1, out_scale=d.to(dev), dotx=dotx.to(dev), in_mul2=mul2)
    assert torch.equal(y2, y3) and torch.equal(dot, dot3)


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 128, 128, 32, 64), (1, 144, 192, 40, 33), (2, 256, 64, 17, 32)])
def test_conv3x3_big_tile_kernel(dev, B, Ci, Co, H, W, tunable):
    """>= 128 -> >= 64 channel stride-1 convs with enough tiles take the 16x32-tile two-stage kernel
    (conv_f16s_big.hip); the item threshold is lowered so that small tensors reach it."""
    import torch.nn.functional as F
    from oodgan import ops
    tunable('s1_big_min_items', 1)
    x = synth.normal('bg.x', (B, Ci, H, W), 1)
    w = synth.normal('bg.w', (Co, Ci, 3, 3), 2, 1.0 / math.sqrt(Ci * 9))
    s = synth.normal('bg.s', (B, Ci), 3, 0.3, 1.0)
    d = synth.normal('bg.d', (B, Co), 4, 0.3, 1.0)
    nz = synth.normal('bg.nz', (B, 1, H, W), 5)
    nw = torch.tensor([0.37])
    bias = synth.normal('bg.b', (Co,), 6)
    xs = ops.to_sform(x.to(dev), s.to(dev))
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    raw = F.conv2d(x * s[:, :, None, None], w, padding=1)
    y = ops.conv3x3(xs, wpk, Co, ops.CONV_S1, out_scale=d.to(dev), bias=bias.to(dev), noise=nz.to(dev), noise_weight=nw.to(dev),
                    act=ops.ACT_LRELU)
    close(y, R.fused_leaky_relu(raw * d[:, :, None, None] + nw * nz, bias))
    mul2 = torch.tensor([2.0 ** -5, 2.0 ** 5], device=dev)
    xs2 = ops.to_sform(x.to(dev), s.to(dev), mul2)
    dotx = synth.normal('bg.dx', (B, Co, H, W), 8)
    y2, dot = ops.conv3x3(xs2, wpk, Co, ops.CONV_S1, out_scale=d.to(dev), dotx=dotx.to(dev), in_mul2=mul2)
    close(y2, raw * d[:, :, None, None])
    close(dot, (raw * dotx).sum(dim=(2, 3)), 2e-4)
    # affine input (the SAMM bottlenecks: InstanceNorm folded into the S-form conversion, zero padding AFTER it) + PReLU epilogue
    sh = synth.normal('bg.sh', (B, Ci), 9, 0.5)
    slope = synth.normal('bg.sl', (Co,), 10, 0.1, 0.25)
    xa = ops.to_sform(x.to(dev), s.to(dev), shift=sh.to(dev))
    y4 = ops.conv3x3(xa, wpk, Co, ops.CONV_S1, act=ops.ACT_PRELU, slope=slope.to(dev))
    ref4 = F.conv2d(x * s[:, :, None, None] + sh[:, :, None, None], w, padding=1)
    close(y4, torch.where(ref4 > 0, ref4, ref4 * slope.view(1, -1, 1, 1)))
    tunable('s1_big_min_items', 1000000000)         # same calls through the tile kernel
    y3, dot3 = ops.conv3x3(xs2, wpk, Co, ops.CONV_S1, out_scale=d.to(dev), dotx=dotx.to(dev), in_mul2=mul2)
    assert (y3 - y2).abs().max().item() <= 1e-5 * y2.abs().max().item()
    y5 = ops.conv3x3(xa, wpk, Co, ops.CONV_S1, act=ops.ACT_PRELU, slope=slope.to(dev))
    assert (y5 - y4).abs().max().item() <= 1e-5 * y4.abs().max().item()


@pytest.mark.parametrize('B,C,H,W', [(2, 32, 16, 32), (1, 64, 72, 128), (2, 16, 6, 4)])
def test_torgb_with_sform_output_matches_separate_passes(dev, B, C, H, W):
    """ToRGB that also emits the next up-conv's S-form input: y as the plain ToRGB (and the oracle), S-form as to_sform."""
    from oodgan import ops
    seed = 31 + C
    x = synth.normal('x', (B, C, H, W), seed)
    w = synth.normal('w', (3, C), seed)
    s = synth.normal('s', (B, C), seed, 0.3, 1.0)
    s_next = synth.normal('s2', (B, C), seed, 0.3, 1.0)
    bias = synth.normal('b', (3,), seed, 0.2)
    skip = synth.normal('skip', (B, 3, H // 2, W // 2), seed)
    k = R.make_kernel([1, 3, 3, 1]) * 4.0
    for sk in (skip, None):
        ys = ops.SForm(B, C, H, W, dev)
        y = ops.torgb(x.to(dev), w.to(dev), s.to(dev), bias.to(dev), None if sk is None else sk.to(dev), k.to(dev) if sk is not None else None,
                      ys=ys, ys_scale=s_next.to(dev))
        y_plain = ops.torgb(x.to(dev), w.to(dev), s.to(dev), bias.to(dev), None if sk is None else sk.to(dev), k.to(dev) if sk is not None else None)
        close(y, y_plain.cpu(), 1e-6)
        ref = torch.einsum('kc,bc,bchw->bkhw', w, s, x) / math.sqrt(C) + bias.view(1, 3, 1, 1)
        if sk is not None:
            ref = ref + R.upfirdn2d(sk, k, up=2, pad=(2, 1))
        close(y, ref)
        # same split-f16 representation as to_sform: identical hi halves, hi+lo equal to fp32 rounding (the lo half may
        # differ in its last bit: packed vs scalar conversion on exact ties)
        a = ys.data.reshape(-1, 4, 8).float()
        b_ = ops.to_sform(x.to(dev), s_next.to(dev)).data.reshape(-1, 4, 8).float()
        assert torch.equal(a[:, :2], b_[:, :2])
        va, vb = a[:, :2] + a[:, 2:], b_[:, :2] + b_[:, 2:]
        assert (va - vb).abs().max().item() <= 2.0 ** -21 * max(1.0, vb.abs().max().item())


@pytest.mark.parametrize('B,C,H,W', [(2, 32, 16, 64), (1, 32, 40, 96), (1, 24, 13, 45)])
def test_strip_conv_with_fused_torgb_matches_separate_torgb(dev, B, C, H, W):
    """The 32-channel strip kernel also forms the three ToRGB colour sums of its activated output (full and ragged tiles);
    oodgan_rgb_finish adds bias and the up-sampled skip: same result as the conv followed by the ToRGB kernel."""
    from oodgan import ops
    seed = 41 + C + H
    x = synth.normal('x', (B, C, H, W), seed)
    w = synth.normal('w', (C, C, 3, 3), seed, 1.0 / math.sqrt(9 * C))
    s = synth.normal('s', (B, C), seed, 0.3, 1.0)
    d = synth.normal('d', (B, C), seed, 0.2, 1.0)
    bias = synth.normal('b', (C,), seed, 0.2)
    noise = synth.normal('nz', (B, 1, H, W), seed)
    nw = torch.tensor([0.3])
    w_rgb = synth.normal('wr', (3, C), seed)
    s_rgb = synth.normal('sr', (B, C), seed, 0.3, 1.0)
    b_rgb = synth.normal('br', (3,), seed, 0.2)
    k = (R.make_kernel([1, 3, 3, 1]) * 4.0).to(dev)
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    xs = ops.to_sform(x.to(dev), s.to(dev))
    kw = dict(out_scale=d.to(dev), bias=bias.to(dev), noise=noise.to(dev), noise_weight=nw.to(dev), act=ops.ACT_LRELU)
    y_ref = ops.conv3x3(xs, wpk, C, ops.CONV_S1, **kw)
    y, part = ops.conv3x3(xs, wpk, C, ops.CONV_S1, rgb=(w_rgb.to(dev), s_rgb.to(dev)), **kw)
    assert torch.equal(y, y_ref)
    skips = [None]
    if H % 2 == 0 and W % 2 == 0:
        skips.append(synth.normal('skip', (B, 3, H // 2, W // 2), seed).to(dev))
    for sk in skips:
        ref = ops.torgb(y_ref, w_rgb.to(dev), s_rgb.to(dev), b_rgb.to(dev), sk, k if sk is not None else None)
        got = ops.rgb_finish(part.clone(), b_rgb.to(dev), sk, k if sk is not None else None)
        close(got, ref.cpu(), 2e-6)


def test_upfirdn2d_down2_4x4_kernel_vs_oracle(dev):
    """the specialised down-2 / 4x4 path (gradient of the ToRGB skip up-sampling), odd and even sizes, asymmetric pads"""
    from oodgan import ops
    k = R.make_kernel([1, 3, 3, 1]) * 4.0
    k[1, 2] += 0.25                                  # non-symmetric: the flip matters
    for (H, W), pad in (((64, 64), (1, 1)), ((37, 50), (1, 1)), ((16, 24), (2, 1)), ((9, 9), (0, 3))):
        x = synth.normal(f'd2.{H}', (2, 3, H, W), 7)
        close(ops.upfirdn2d(x.to(dev), k.to(dev), 1, 2, pad), R.upfirdn2d(x, k, 1, 2, pad))


@pytest.mark.parametrize('B,Co,Ci,H,W', [(2, 64, 128, 16, 32), (1, 48, 256, 9, 40), (2, 32, 64, 24, 33), (1, 80, 192, 8, 64)])
def test_conv3x3_s2_big_kernel(dev, B, Co, Ci, H, W, tunable):
    """Stride-2 conv (input gradient of the up-sampling conv: K = Co channels of the gradient, M = Ci) on the phase-split
    S-form through conv_f16s_s2big.hip — both instances (128 / 64 channels per workgroup), ragged tiles, with the
    style-gradient dot and a power-of-two input range scale — against autograd of conv_transpose2d and against the
    two-group tile kernel it replaces."""
    import torch.nn.functional as F
    from oodgan import ops
    tunable('s2_big_min_items', 0)
    x = synth.normal('s2b.x', (B, Ci, H, W), 1)
    w = synth.normal('s2b.w', (Co, Ci, 3, 3), 2, 1.0 / math.sqrt(Ci * 9))
    s = synth.normal('s2b.s', (B, Ci), 3, 0.3, 1.0)
    d = synth.normal('s2b.d', (B, Co), 4, 0.3, 1.0)
    gz = synth.normal('s2b.gz', (B, Co, 2 * H + 1, 2 * W + 1), 5)
    xs_ = (x * s[:, :, None, None]).clone().requires_grad_(True)
    (F.conv_transpose2d(xs_, w.transpose(0, 1), stride=2) * d[:, :, None, None] * gz).sum().backward()
    wpk_t = ops.pack_conv3x3(w.to(dev), 1.0, transpose=True, flip=False, precision='f16s')
    mul2 = torch.tensor([2.0 ** -4, 2.0 ** 4], device=dev)
    gp = ops.to_sform_phases(gz.to(dev), H, W, d.to(dev), mul2)
    dx, dot = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s.to(dev), dotx=x.to(dev), in_mul2=mul2)
    close(dx, xs_.grad * s[:, :, None, None], 2e-4)
    close(dot, (xs_.grad * x).sum(dim=(2, 3)), 2e-4)
    dx1 = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s.to(dev), in_mul2=mul2)          # no dot epilogue
    assert torch.equal(dx1, dx)
    dx2, dot2 = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s.to(dev), dotx=x.to(dev), in_mul2=mul2)
    assert torch.equal(dx2, dx) and torch.equal(dot2, dot)                                      # deterministic
    tunable('s2_big_min_items', 1000000000)                                 # the tile kernel
    dx3, dot3 = ops.conv3x3(gp, wpk_t, Ci, ops.CONV_S2, out_scale=s.to(dev), dotx=x.to(dev), in_mul2=mul2)
    assert (dx3 - dx).abs().max().item() <= 1e-5 * dx.abs().max().item()
    assert (dot3 - dot).abs().max().item() <= 1e-4 * dot.abs().max().item()


@pytest.mark.parametrize('B,G,K,Mg,H,W', [(2, 3, 64, 64, 8, 8), (1, 5, 48, 128, 4, 6), (3, 2, 32, 64, 2, 2)])
def test_conv3x3_s2_grouped(dev, B, G, K, Mg, H, W):
    """oodgan_conv_args.groups: nn.Conv2d(G*K, G*Mg, 3, stride 2, padding 1, groups=G) — the style heads of the e4e encoder
    advancing side by side (psp_encoders.py:14-34) — with bias and per-channel slopes, down to 1x1 outputs."""
    import torch.nn.functional as F
    from oodgan import ops
    x = synth.normal('gr.x', (B, G * K, H, W), 1)
    w = synth.normal('gr.w', (G * Mg, K, 3, 3), 2, 1.0 / math.sqrt(K * 9))
    bias = synth.normal('gr.b', (G * Mg,), 3)
    slope = torch.full((G * Mg,), 0.01)
    ref = F.leaky_relu(F.conv2d(x, w, bias, stride=2, padding=1, groups=G), 0.01)
    pitch = (W + 1 + 3) // 4 * 4
    xp = torch.zeros(B, G * K, H + 1, pitch)
    xp[:, :, 1:, 1:W + 1] = x
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    y = ops.conv3x3(xp.to(dev), wpk, G * Mg, ops.CONV_S2, in_hw=(H + 1, W + 1), in_pitch=pitch, bias=bias.to(dev), act=ops.ACT_PRELU,
                    slope=slope.to(dev), groups=G)
    close(y, ref, 2e-5)
    with pytest.raises(RuntimeError):           # Mg must be a multiple of 64: a channel block may not straddle two groups
        ops.conv3x3(xp.to(dev), wpk, G * Mg, ops.CONV_S2, in_hw=(H + 1, W + 1), in_pitch=pitch, groups=G * 4)


@pytest.mark.parametrize('B,G,K,Mg,H,W', [(2, 3, 64, 128, 20, 20), (1, 5, 48, 128, 8, 20), (2, 2, 32, 256, 34, 32)])
def test_conv3x3_s2_grouped_sform(dev, B, G, K, Mg, H, W, tunable):
    """The grouped stride-2 conv on the 8-wave kernel (round 4): phase-split S-form of all G*K input channels, every channel block
    of the kernel reading its own group's K channels.  Same result as the fp32-input grouped kernel to rounding, dispatch counted."""
    import torch.nn.functional as F
    from oodgan import _lib, ops
    tunable('s2_big_min_items', 0)
    x = synth.normal('grs.x', (B, G * K, H, W), 1, 20.0)
    w = synth.normal('grs.w', (G * Mg, K, 3, 3), 2, 1.0 / math.sqrt(K * 9))
    bias = synth.normal('grs.b', (G * Mg,), 3)
    slope = synth.normal('grs.sl', (G * Mg,), 4, 0.05, 0.1)
    ref = F.prelu(F.conv2d(x, w, bias, stride=2, padding=1, groups=G), slope)
    pitch = (W + 1 + 3) // 4 * 4
    xp = torch.zeros(B, G * K, H + 1, pitch)
    xp[:, :, 1:, 1:W + 1] = x
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    assert ops.s2_grouped_supported(B, K, G * Mg, G, H + 1, W + 1)
    assert not ops.s2_grouped_supported(B, K, G * 64, G, H + 1, W + 1)            # Mg % 128
    mul2 = ops.absmax_mul2(x.to(dev))
    gp = ops.to_sform_phases(xp.to(dev), H // 2, W // 2, mul2=mul2, in_pitch=pitch)
    gp2 = ops.to_sform_phases(x.to(dev), H // 2, W // 2, mul2=mul2, pad_tl=True)       # the zero pad made by the conversion itself
    assert torch.equal(gp.data, gp2.data)
    _lib.dispatch_reset()
    y = ops.conv3x3(gp, wpk, G * Mg, ops.CONV_S2, bias=bias.to(dev), in_mul2=mul2, act=ops.ACT_PRELU, slope=slope.to(dev), groups=G)
    assert _lib.dispatch_count('s2big') == 1
    close(y, ref, 2e-5)
    y2 = ops.conv3x3(xp.to(dev), wpk, G * Mg, ops.CONV_S2, in_hw=(H + 1, W + 1), in_pitch=pitch, bias=bias.to(dev), act=ops.ACT_PRELU,
                     slope=slope.to(dev), groups=G)
    assert (y - y2).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize('B,K,M,H,W,both', [(2, 64, 64, 16, 32, True), (1, 128, 96, 20, 40, False), (2, 64, 80, 9, 33, True), (1, 64, 128, 32, 64, False)])
def test_conv3x3_s1_big_kernel_sform_output(dev, B, K, M, H, W, both, tunable):
    """Round 4: the 8-wave stride-1 kernel writes the NEXT conv's S-form input from its registers (oodgan_conv_args.ys; y optional) —
    AlignNet's conv -> PReLU -> conv (SAMM/helpers.py:96-109 through e4e helpers.py:426-448) without the fp32 tensor in between.
    Bit-identical to converting the kernel's own fp32 output; the chained second conv against torch."""
    import torch.nn.functional as F
    from oodgan import _lib, ops
    tunable('s1_big_min_items', 0)
    assert ops.s1_ys_supported(B, K, M, H, W)
    x = synth.normal('ys.x', (B, K, H, W), 1)
    w1 = synth.normal('ys.w1', (M, K, 3, 3), 2, 1.0 / math.sqrt(K * 9))
    w2 = synth.normal('ys.w2', (M, M, 3, 3), 3, 1.0 / math.sqrt(M * 9))
    slope = synth.normal('ys.sl', (M,), 4, 0.05, 0.1)
    bias = synth.normal('ys.b', (M,), 5)
    e = 9
    ysc = torch.full((B, M), 2.0 ** e)
    ysc[:, ::3] *= 2.0                                        # per-channel scales are honoured (the second conv un-scales by hand below)
    mul2 = torch.tensor([2.0 ** -e, 2.0 ** e])
    xs = ops.to_sform(x.to(dev))
    p1, p2 = ops.pack_conv3x3(w1.to(dev), precision='f16s'), ops.pack_conv3x3(w2.to(dev), precision='f16s')
    kw = dict(bias=bias.to(dev), act=ops.ACT_PRELU, slope=slope.to(dev))
    y = ops.conv3x3(xs, p1, M, ops.CONV_S1, **kw)                                  # plain instance
    ref1 = F.prelu(F.conv2d(x, w1, bias, padding=1), slope)
    close(y, ref1, 2e-5)
    want = ops.to_sform(y, ysc.to(dev))
    ys = ops.SForm(B, M, H, W, dev)
    _lib.dispatch_reset()
    out = ops.conv3x3(xs, p1, M, ops.CONV_S1, ys=ys, ys_scale=ysc.to(dev), want_y=both, **kw)
    assert _lib.dispatch_count('s1big') == 1
    assert torch.equal(ys.data, want.data)
    if both:
        assert torch.equal(out, y)
    else:
        assert out is None
    # fused ToRGB partial sums (one per 64-channel block, ModulatedConv2d 1x1 without demodulation: model.py:363-372) and the maximum
    # of what went into the S-form (forward range control of its reader), from the same launch
    w_rgb = synth.normal('ys.wr', (3, M), 6)
    s_rgb = synth.normal('ys.sr', (B, M), 7, 0.3, 1.0)
    vm = torch.zeros(B * ops.VMAX_SLOTS, device=dev, dtype=torch.int32)
    out2, part = ops.conv3x3(xs, p1, M, ops.CONV_S1, ys=ys, ys_scale=ysc.to(dev), rgb=(w_rgb.to(dev), s_rgb.to(dev)), vmax=vm, **kw)
    assert torch.equal(out2, y) and torch.equal(ys.data, want.data)
    rgb = ops.rgb_finish(part.clone())
    ref_rgb = torch.einsum('km,bm,bmhw->bkhw', w_rgb, s_rgb, ref1) / math.sqrt(M)
    close(rgb, ref_rgb, 2e-5)
    got = vm.view(torch.float32).reshape(B, -1).max(dim=1).values.cpu()
    exp = (y * ysc.to(dev)[:, :, None, None]).abs().amax(dim=(1, 2, 3)).cpu()
    assert torch.equal(got, exp)
    # the chain: second conv on the S-form written by the first (uniform scale here, undone by in_mul2)
    ysu = torch.full((B, M), 2.0 ** e)
    ops.conv3x3(xs, p1, M, ops.CONV_S1, ys=ys, ys_scale=ysu.to(dev), want_y=False, **kw)
    z = ops.conv3x3(ys, p2, M, ops.CONV_S1, in_mul2=mul2.to(dev))
    close(z, F.conv2d(ref1, w2, padding=1), 2e-5)


@pytest.mark.parametrize('B,K,M,H,W,act', [(2, 64, 128, 16, 32, 'prelu'), (1, 48, 192, 10, 40, 'lrelu'), (2, 32, 64, 24, 34, 'none')])
def test_conv3x3_s2_big_kernel_forward_use(dev, B, K, M, H, W, act, tunable):
    """The 8-wave stride-2 kernel as a FORWARD conv (nn.Conv2d(K, M, 3, stride 2, padding 1) of the e4e encoder's
    GradualStyleBlocks, psp_encoders.py:14-34): input padded by one zero row / column on the top / left, phase-split S-form
    with a power-of-two range scale, bias + PReLU / leaky-ReLU*sqrt2 in the epilogue; both channel-block instances."""
    import torch.nn.functional as F
    from oodgan import ops
    tunable('s2_big_min_items', 0)
    x = synth.normal('s2f.x', (B, K, H, W), 1, 30.0)
    w = synth.normal('s2f.w', (M, K, 3, 3), 2, 1.0 / math.sqrt(K * 9))
    bias = synth.normal('s2f.b', (M,), 3)
    slope = synth.normal('s2f.sl', (M,), 4, 0.05, 0.1)
    ref = F.conv2d(x, w, bias, stride=2, padding=1)
    ref = {'prelu': F.prelu(ref, slope), 'lrelu': F.leaky_relu(ref, 0.2) * math.sqrt(2.0), 'none': ref}[act]
    pitch = (W + 1 + 3) // 4 * 4
    xp = torch.zeros(B, K, H + 1, pitch)
    xp[:, :, 1:, 1:W + 1] = x
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    mul2 = ops.absmax_mul2(x.to(dev))
    assert 512.0 <= x.abs().max().item() * mul2[1].item() < 1024.0
    gp = ops.to_sform_phases(xp.to(dev), H // 2, W // 2, mul2=mul2, in_pitch=pitch)
    kw = {'prelu': dict(act=ops.ACT_PRELU, slope=slope.to(dev)), 'lrelu': dict(act=ops.ACT_LRELU), 'none': {}}[act]
    y = ops.conv3x3(gp, wpk, M, ops.CONV_S2, bias=bias.to(dev), in_mul2=mul2, **kw)
    close(y, ref, 2e-5)


@pytest.mark.parametrize('B,Ci,Co,H,W', [(2, 64, 64, 16, 32), (1, 48, 128, 9, 40), (2, 32, 96, 23, 31), (1, 80, 64, 8, 64)])
def test_conv3x3_t2_big_kernel(dev, B, Ci, Co, H, W, tunable):
    """Transposed stride-2 conv (the up-sampling ModulatedConv2d before its blur) from an S-form input through
    conv_f16s_t2big.hip — ragged position grids, partial channel blocks — against conv_transpose2d and against the
    4-wave kernel it replaces."""
    import torch.nn.functional as F
    from oodgan import ops
    tunable('t2_big_min_items', 0)
    x = synth.normal('t2b.x', (B, Ci, H, W), 1)
    w = synth.normal('t2b.w', (Co, Ci, 3, 3), 2, 1.0 / math.sqrt(Ci * 9))
    s = synth.normal('t2b.s', (B, Ci), 3, 0.3, 1.0)
    d = synth.normal('t2b.d', (B, Co), 4, 0.3, 1.0)
    xs = ops.to_sform(x.to(dev), s.to(dev))
    wpk = ops.pack_conv3x3(w.to(dev), precision='f16s')
    ref = F.conv_transpose2d(x * s[:, :, None, None], w.transpose(0, 1), stride=2) * d[:, :, None, None]
    z = ops.conv3x3(xs, wpk, Co, ops.CONV_T2, out_scale=d.to(dev))
    close(z[..., :2 * W + 1], ref, 2e-4)
    z2 = ops.conv3x3(xs, wpk, Co, ops.CONV_T2, out_scale=d.to(dev))
    assert torch.equal(z2[..., :2 * W + 1], z[..., :2 * W + 1])
    tunable('t2_big_min_items', 1000000000)
    z3 = ops.conv3x3(xs, wpk, Co, ops.CONV_T2, out_scale=d.to(dev))
    assert (z3[..., :2 * W + 1] - z[..., :2 * W + 1]).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_zero_fill_kernel_and_grouped_equal_linear(dev):
    """Round 4 helpers: oodgan_zero (a fill kernel: unaligned starts / odd byte counts leave the neighbours alone) and
    oodgan_equal_linear_grouped (G EqualLinear layers side by side, bit-identical to G oodgan_equal_linear calls — the final linears of the
    e4e style heads, psp_encoders.py:31-34)."""
    from oodgan import _lib, ops
    buf = torch.full((4099,), 7, device=dev, dtype=torch.uint8)
    for (off, n) in ((0, 4099), (1, 33), (5, 4000), (16, 16), (3, 1), (17, 0)):
        buf.fill_(7)
        rc = _lib.lib().oodgan_zero(ctypes.c_void_p(buf.data_ptr() + off), n, ops._stream())
        assert rc == 0
        ref = torch.full((4099,), 7, dtype=torch.uint8)
        ref[off:off + n] = 0
        assert torch.equal(buf.cpu(), ref), (off, n)
    z = ops.zeros(3, 5, 7, device=dev)
    assert z.shape == (3, 5, 7) and float(z.abs().max()) == 0.0
    B, G, I, O = 3, 5, 96, 64
    x = synth.normal('gl.x', (B, G, I), 1).to(dev)
    w = synth.normal('gl.w', (G, O, I), 2).to(dev)
    b = synth.normal('gl.b', (G, O), 3).to(dev)
    for lr_mul, act in ((1.0, False), (0.01, True)):
        y = ops.equal_linear_grouped(x, w, b, lr_mul=lr_mul, activation=act)
        for g in range(G):
            yg = ops.equal_linear(x[:, g].contiguous(), w[g], b[g], lr_mul=lr_mul, activation=act)
            assert torch.equal(y[:, g], yg)
