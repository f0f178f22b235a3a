"""fp16 modulated conv of the high-resolution layers (BASELINE.json configs[4]) against the oracle's
ModulatedConv2d + NoiseInjection + FusedLeakyReLU on the same f16-rounded input.

Tolerance: the kernel rounds the per-sample weights and the result to f16 (unit round-off 2^-11 = 4.9e-4) and
accumulates in fp32, so |diff| <= 4e-3 * max|ref| element-wise and <= 5e-4 * max|ref| on average."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_cpu as R  # noqa: E402
from oodgan import synth  # noqa: E402


def _case(B, K, M, H, W, seed, noise_batch, act):
    x = synth.normal('x', (B, K, H, W), seed).half().float()
    wgt = synth.normal('w', (1, M, K, 3, 3), seed)
    s = synth.normal('s', (B, K), seed, 0.3, 1.0)
    noise = synth.normal('n', (noise_batch, 1, H, W), seed) if noise_batch else None
    bias = synth.normal('b', (M,), seed, 0.2)
    nw = torch.tensor([0.37])
    y = R.modulated_conv2d(x, s, wgt, torch.eye(K) * math.sqrt(K), torch.zeros(K))
    if noise is not None:
        y = y + nw * noise
    y = R.fused_leaky_relu(y, bias) if act == 'lrelu' else y + bias.reshape(1, -1, 1, 1)
    return x, wgt, s, noise, bias, nw, y


@pytest.mark.parametrize('B,K,M,H,W,noise_batch,act', [
    (2, 32, 32, 16, 32, 2, 'lrelu'),      # exact tiles
    (3, 32, 32, 24, 64, 1, 'lrelu'),      # shared noise, several tiles per sample
    (2, 16, 32, 13, 45, 2, 'none'),       # ragged tile edges, one channel block in
    (1, 32, 16, 9, 33, 0, 'lrelu'),       # one channel block out, no noise
    (2, 24, 20, 40, 70, 2, 'lrelu'),      # channel counts that are not multiples of 16
])
def test_modconv_f16_vs_oracle(B, K, M, H, W, noise_batch, act):
    from oodgan import ops
    dev = torch.device('cuda:0')
    x, wgt, s, noise, bias, nw, ref = _case(B, K, M, H, W, 7 + B + H, noise_batch, act)
    xh = ops.to_hform(x.to(dev))
    assert torch.equal(xh.to_nchw().cpu(), x)                     # layout round trip is exact
    packed = ops.modconv_f16_pack(wgt.to(dev), s.to(dev), act=act)
    out = ops.modconv_f16(xh, packed, None if noise is None else noise.to(dev), nw.to(dev), bias.to(dev))
    y = out.to_nchw().cpu()
    scale = max(1.0, ref.abs().max().item())
    err = (y - ref).abs()
    assert err.max().item() <= 4e-3 * scale, err.max().item()
    assert err.mean().item() <= 5e-4 * scale, err.mean().item()
    # the border / padding of the output must stay zero (the next conv reads it as its halo)
    full = out.buf.clone()
    ops.to_hform(torch.zeros_like(out.to_nchw()), out=out)
    assert out.buf.abs().max().item() == 0.0
    del full


def test_modconv_f16_pack_with_the_affine_inside():
    """oodgan_modconv_f16_pack_affine: the modulation EqualLinear (model.py:219-223,236) inside the pack kernel gives the packed
    weights of the two-launch form (style_affine, then pack) up to the f16 rounding of a style that differs in its last fp32 bit."""
    from oodgan import ops
    dev = torch.device('cuda:0')
    B, K, M, S = 5, 32, 32, 512
    lat = synth.normal('pa.l', (B, S), 1).to(dev)
    mw = synth.normal('pa.w', (K, S), 2).to(dev)
    mb = synth.normal('pa.b', (K,), 3, 0.1, 1.0).to(dev)
    wgt = synth.normal('pa.c', (M, K, 3, 3), 4).to(dev)
    s = ops.style_affine(lat, mw, mb)
    ref = (lat.double() @ mw.double().t() / math.sqrt(S) + mb.double()).float()
    assert (s - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    a = ops.modconv_f16_pack(wgt, s, act='lrelu')[0].float()
    b = ops.modconv_f16_pack(wgt, None, act='lrelu', latent=lat, mod_weight=mw, mod_bias=mb)[0].float()
    assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-3 * a.abs().max().item()
    assert ((a - b).abs() > 0).float().mean().item() < 0.05          # f16 values: almost all identical
    again = ops.modconv_f16_pack(wgt, None, act='lrelu', latent=lat, mod_weight=mw, mod_bias=mb)[0].float()
    assert torch.equal(again, b)


def test_modconv_f16_is_deterministic_and_persistent_grid_covers_all_tiles():
    """many more tiles than resident workgroups: every tile written exactly once, run-to-run identical."""
    from oodgan import ops
    dev = torch.device('cuda:0')
    B, K, M, H, W = 2, 32, 32, 512, 512
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, K, H, W, generator=g).half().float()
    wgt = torch.randn(1, M, K, 3, 3, generator=g)
    s = 1 + 0.3 * torch.randn(B, K, generator=g)
    xh = ops.to_hform(x.to(dev))
    packed = ops.modconv_f16_pack(wgt.to(dev), s.to(dev))
    y1 = ops.modconv_f16(xh, packed).to_nchw()
    y2 = ops.modconv_f16(xh, packed).to_nchw()
    assert torch.equal(y1, y2)
    ref = R.modulated_conv2d(x[:, :, :64, :96], s, wgt, torch.eye(K) * math.sqrt(K), torch.zeros(K))
    # interior of the crop (its last row/col see different neighbours than the full image)
    err = (y1[:, :, :63, :95].cpu() - ref[:, :, :63, :95]).abs().max().item()
    assert err <= 4e-3 * max(1.0, ref.abs().max().item())
    # linearity in the activations (size-independent property): conv(2x) == 2 conv(x) exactly in f16/f32 arithmetic
    x2 = ops.to_hform((2 * x).to(dev))
    y3 = ops.modconv_f16(x2, packed).to_nchw()
    assert (y3 - 2 * y1).abs().max().item() <= 2.0 ** -23      # equal up to f16 subnormal rounding of tiny outputs


def test_modconv_f16_at_the_benchmarked_size_crops_vs_oracle():
    """BASELINE configs[4] itself — 32 -> 32 channels, 1024x1024, batch 16 (what bench.py's M2 leg times): crops of three samples
    (a corner with the image border, the centre, the far corner) against the oracle evaluated on the crop plus its halo."""
    from oodgan import ops
    dev = torch.device('cuda:0')
    B, K, M, H, W = 16, 32, 32, 1024, 1024
    g = torch.Generator().manual_seed(11)
    wgt = torch.randn(1, M, K, 3, 3, generator=g)
    s = 1 + 0.3 * torch.randn(B, K, generator=g)
    bias = 0.1 * torch.randn(M, generator=g)
    nw = torch.tensor([0.2])
    xd = torch.empty(B, K, H, W, device=dev)
    for b in range(B):                          # generated per sample: no 2 GB host tensor
        xd[b] = torch.randn(K, H, W, generator=g).half().float().to(dev)
    noise = torch.randn(B, 1, H, W, generator=g).to(dev)
    xh = ops.to_hform(xd)
    packed = ops.modconv_f16_pack(wgt.to(dev), s.to(dev), act='lrelu')
    y = ops.modconv_f16(xh, packed, noise, nw.to(dev), bias.to(dev)).to_nchw()
    assert torch.isfinite(y).all()
    for b, (y0, x0) in ((0, (0, 0)), (7, (480, 500)), (15, (H - 64, W - 96))):
        ya, xa = max(y0 - 1, 0), max(x0 - 1, 0)
        yb, xb = min(y0 + 65, H), min(x0 + 97, W)
        xc = xd[b:b + 1, :, ya:yb, xa:xb].cpu()
        ref = R.modulated_conv2d(xc, s[b:b + 1], wgt, torch.eye(K) * math.sqrt(K), torch.zeros(K))
        ref = R.fused_leaky_relu(ref + nw * noise[b:b + 1, :, ya:yb, xa:xb].cpu(), bias)
        # rows / columns of the crop whose 3x3 neighbourhood lies inside the crop (or is the true image border)
        r0, c0 = (0 if ya == 0 else 1), (0 if xa == 0 else 1)
        r1, c1 = ref.shape[2] - (0 if yb == H else 1), ref.shape[3] - (0 if xb == W else 1)
        got = y[b:b + 1, :, ya + r0:ya + r1, xa + c0:xa + c1].cpu()
        err = (got - ref[:, :, r0:r1, c0:c1]).abs().max().item()
        assert err <= 4e-3 * max(1.0, ref.abs().max().item()), (b, err)
