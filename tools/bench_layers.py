#!/usr/bin/env python3
"""Per-layer timing of the matrix kernels at the PRODUCTION shapes of the inversion loop (StyleGAN2-1024, B=8): every
styled conv's forward (S1 / T2 on S-form input) and input gradient (S1 / S2 with the style-gradient dot), HIP events
around 10 launches each.  `python tools/bench_layers.py [S1,T2,S2] [B]`"""
import math
import os
import sys

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import ops  # noqa: E402

dev = torch.device('cuda:0')
ops.USE_TINY = os.environ.get('OODGAN_NO_TINY') is None       # the skinny-GEMM kernel of the 4x4 / 8x8 layers
which = sys.argv[1].split(',') if len(sys.argv) > 1 else ['S1', 'T2', 'S2']
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
only = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else None      # resolutions
CH = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def report(tag, ms, flops, byts):
    print(f'{tag:34s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s  {byts / ms / 1e6:7.1f} GB/s', flush=True)


def to_fform(x):
    B_, C_, H_, W_ = x.shape
    return ops.FForm(x.view(B_, C_ // 16, 16, H_, W_).permute(0, 1, 3, 4, 2).contiguous().view(B_, C_, H_, W_))


if 'SX' in which:               # the 1024² level inside the W+ loop: two-pass path against the strip conv that converts its input itself
    C, H = 32, 1024
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
    wf = ops.pack_conv3x3(w, precision='f16s')
    wb = ops.pack_conv3x3(w, transpose=True, flip=True, precision='f16s')
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    d = (1 + 0.3 * torch.randn(B, C, generator=g)).abs().to(dev)
    nz = torch.randn(B, 1, H, H, generator=g).to(dev)
    nw, bias = torch.tensor([0.1], device=dev), torch.zeros(C, device=dev)
    w_rgb, s_rgb = torch.randn(3, C, generator=g).to(dev), (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    g_rgb = (1e-3 * torch.randn(B, 3, H, H, generator=g)).to(dev)
    mul2 = torch.tensor([2.0 ** -14, 2.0 ** 14], device=dev)
    xs, xf = ops.to_sform(x, s), to_fform(x)
    out2 = torch.randn(B, C, H, H, generator=g).to(dev)
    o2f = to_fform(out2)
    del out2
    fl, by = 2.0 * B * C * C * 9 * H * H, 4.0 * B * C * H * H
    report('fwd S-form in, F-form out, rgb', timeit(lambda: ops.conv3x3(xs, wf, C, ops.CONV_S1, out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU, rgb=(w_rgb, s_rgb), y_fform=True)), fl, 2 * by)
    report('fwd F-form in (x_fform 1), rgb', timeit(lambda: ops.conv3x3(xf, wf, C, ops.CONV_S1, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU, rgb=(w_rgb, s_rgb))), fl, 2 * by)
    gin = ops.SForm(B, C, H, H, dev)
    report('bwd producer act_bwd_sform_f', timeit(lambda: ops.act_bwd_producer(o2f, None, nz, nw, bias, d, mul2, gin, g_rgb=g_rgb, w_rgb=w_rgb, s_rgb=s_rgb)), 0, 2 * by)
    report('bwd strip conv + dot + act grad', timeit(lambda: ops.conv3x3(gin, wb, C, ops.CONV_S1, out_scale=s, dotx=x, in_mul2=mul2, dot_actgrad=ops.DotActGrad())), fl, 3 * by)
    report('bwd x_fform 2 (both in one)', timeit(lambda: ops.conv3x3(o2f, wb, C, ops.CONV_S1, out_scale=s, dotx=xf, in_mul2=mul2, dot_actgrad=ops.DotActGrad(),
                                                                     xf_act=ops.ActBwdX(nz, nw, bias, d, mul2, g_rgb, w_rgb, s_rgb))), fl, 3 * by)
    sys.exit(0)

for res in (4, 8, 16, 32, 64, 128, 256, 512, 1024):
    if only and res not in only:
        continue
    cin, cout = CH[max(res // 2, 4)], CH[res]
    g = torch.Generator().manual_seed(res)
    if 'S1' in which:           # plain conv at `res`: cout -> cout
        C, H = cout, res
        w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
        x = torch.randn(B, C, H, H, generator=g).to(dev)
        s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
        xs = ops.to_sform(x, s)
        wf = ops.pack_conv3x3(w, precision='f16s')
        wb = ops.pack_conv3x3(w, transpose=True, flip=True, precision='f16s')
        nz = torch.randn(B, 1, H, H, generator=g).to(dev)
        bias = torch.zeros(C, device=dev)
        nw = torch.tensor([0.1], device=dev)
        fl, by = 2.0 * B * C * C * 9 * H * H, 4.0 * B * 2 * C * H * H
        report(f'S1 fwd {C}->{C} @{H}', timeit(lambda: ops.conv3x3(xs, wf, C, ops.CONV_S1, out_scale=s, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU)), fl, by)
        report(f'S1 bwd+dot {C}->{C} @{H}', timeit(lambda: ops.conv3x3(xs, wb, C, ops.CONV_S1, out_scale=s, dotx=x)), fl, by * 1.5)
        wb2 = ops.pack_conv3x3(w, transpose=True, flip=True, precision='f16s')
        wb2.x_hi_only = True            # precision 'f16s-g2': two matrix instructions per product
        report(f'S1 bwd+dot G2 {C}->{C} @{H}', timeit(lambda: ops.conv3x3(xs, wb2, C, ops.CONV_S1, out_scale=s, dotx=x)), fl, by * 1.5)
        del x, xs, nz
    if res == 4:
        continue
    if 'T2' in which:           # up conv res/2 -> res: cin -> cout
        H = res // 2
        w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev)
        x = torch.randn(B, cin, H, H, generator=g).to(dev)
        s = (1 + 0.3 * torch.randn(B, cin, generator=g)).to(dev)
        d = (1 + 0.3 * torch.randn(B, cout, generator=g)).to(dev)
        xs = ops.to_sform(x, s)
        wf = ops.pack_conv3x3(w, precision='f16s')
        fl = 2.0 * B * cin * cout * 9 * H * H
        report(f'T2 fwd {cin}->{cout} @{H}->{res}', timeit(lambda: ops.conv3x3(xs, wf, cout, ops.CONV_T2, out_scale=d)), fl, 4.0 * B * (cin * H * H + cout * (2 * H + 1) ** 2))
        del x, xs
    if 'S2' in which:           # input gradient of the up conv: g (cout @ 2H+1) -> dx (cin @ H)
        H = res // 2
        w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev)
        Hin = 2 * H + 1
        P2 = (Hin + 3) // 4 * 4
        g2 = torch.randn(B, cout, Hin, P2, generator=g).to(dev)
        d = (1 + 0.3 * torch.randn(B, cout, generator=g)).to(dev)
        s = (1 + 0.3 * torch.randn(B, cin, generator=g)).to(dev)
        x = torch.randn(B, cin, H, H, generator=g).to(dev)
        gp = ops.to_sform_phases(g2, H, H, d, in_pitch=P2)
        del g2
        wb = ops.pack_conv3x3(w, transpose=True, flip=False, precision='f16s')
        fl = 2.0 * B * cin * cout * 9 * H * H
        report(f'S2 bwd+dot {cout}->{cin} @{res}->{H}', timeit(lambda: ops.conv3x3(gp, wb, cin, ops.CONV_S2, out_scale=s, dotx=x)), fl, 4.0 * B * (cout * Hin * Hin + 2 * cin * H * H))
        if ops.s2_fuse_supported(B, cout, cin, Hin, Hin):
            # the same conv with the activation backward of the conv layer below in its epilogue (the form the W+ loop runs)
            nz = torch.randn(B, 1, H, H, generator=g).to(dev)
            grgb = (1e-3 * torch.randn(B, 3, H, H, generator=g)).to(dev)
            wrgb, srgb = torch.randn(3, cin, generator=g).to(dev), (1 + 0.3 * torch.randn(B, cin, generator=g)).to(dev)
            dl = (1 + 0.3 * torch.randn(B, cin, generator=g)).abs().to(dev)
            mul2 = torch.tensor([2.0 ** -9, 2.0 ** 9], device=dev)
            dst = ops.SForm(B, cin, H, H, dev)

            def fused():
                fz = ops.ActBwdFusion(dst, nz, torch.tensor([0.1], device=dev), torch.zeros(cin, device=dev), dl, mul2, g_rgb=grgb, w_rgb=wrgb, s_rgb=srgb)
                ops.conv3x3(gp, wb, cin, ops.CONV_S2, out_scale=s, dotx=x, fuse=fz, want_y=False)
            report(f'S2 bwd+dot+actbwd {cout}->{cin} @{res}->{H}', timeit(fused), fl, 4.0 * B * (cout * Hin * Hin + 2 * cin * H * H))
        del gp, x
