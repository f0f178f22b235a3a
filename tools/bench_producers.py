#!/usr/bin/env python3
"""Timing of the layout producers at the production shapes of the inversion loop (B=8): the backward producer of the
up-sampling layers (activation gradient + blur^T + phase split, `act_bwd_blurT`), HIP events around 10 launches.
`OODGAN_BLURT_STRIP=0` selects the tile kernel.  Algorithmic bytes: read 2 tensors (2H x 2W), write 1 (4 B per element).
Second block: the forward tail of the same layers (blur + noise + bias + lrelu + S-form of the next conv, `blur_act_sform`): read z (4 B), write y (4 B) and the S-form (4 B) per element."""
import os
import sys

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import ops  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CH = {64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}


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


k1 = torch.tensor([1., 3., 3., 1.])
k = (k1[:, None] * k1[None, :] / 64 * 4).flip(0, 1).contiguous().to(dev)
for res in (64, 128, 256, 512, 1024):
    C, H = CH[res], res // 2
    g = torch.Generator().manual_seed(res)
    out = torch.randn(B, C, res, res, generator=g).to(dev)
    gf = (1e-3 * torch.randn(B, C, res, res, generator=g)).to(dev)
    nz = torch.randn(B, 1, res, res, generator=g).to(dev)
    nw, bias = torch.tensor([0.1], device=dev), torch.zeros(C, device=dev)
    d = (1 + 0.3 * torch.randn(B, C, generator=g)).abs().to(dev)
    mul2 = torch.tensor([2.0 ** -9, 2.0 ** 9], device=dev)
    dst = ops.SFormPhases(B, C, H, H, dev)
    ms = timeit(lambda: ops.act_bwd_producer(out, gf, nz, nw, bias, d, mul2, dst, blur_kernel=k))
    byts = 4.0 * B * C * (2 * res * res + (res + 1) ** 2)
    print(f'act_bwd_blurT {C:3d} ch @{res:4d}: {ms * 1e3:8.1f} us  {byts / ms / 1e6:7.1f} GB/s', flush=True)
    # the form the loop uses: the conv above already applied act' (one tensor in)
    dg = ops.DotActGrad()
    dg.dot_part, dg.scale = torch.zeros(B, C, 4, device=dev), d
    ms = timeit(lambda: ops.act_bwd_producer(None, gf, nz, nw, bias, d, mul2, dst, blur_kernel=k, dot_of=dg))
    byts = 4.0 * B * C * (res * res + (res + 1) ** 2)
    print(f'act_bwd_blurT {C:3d} ch @{res:4d} pre-activated: {ms * 1e3:8.1f} us  {byts / ms / 1e6:7.1f} GB/s', flush=True)
    del out, gf, nz, dst

kf = (k1[:, None] * k1[None, :] / 64 * 4).contiguous().to(dev)
for res in (64, 128, 256, 512, 1024):
    C, H = CH[res], res // 2
    g = torch.Generator().manual_seed(res)
    pitch = (res + 1 + 3) // 4 * 4
    z = torch.randn(B, C, res + 1, pitch, generator=g).to(dev)
    nz = torch.randn(B, 1, res, res, generator=g).to(dev)
    nw, bias = torch.tensor([0.1], device=dev), torch.zeros(C, device=dev)
    s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    ys = ops.SForm(B, C, res, res, dev)
    vm = torch.zeros(B, ops.VMAX_SLOTS, dtype=torch.int32, device=dev)
    R1 = os.environ.get("OODGAN_BENCH_R1") is not None
    ms = timeit(lambda: ops.blur_act_sform(z, kf, H, H, bias, nz, nw, act=True, ys=ys, ys_scale=s, vmax=vm, rank_one=R1))
    byts = 4.0 * B * C * (2 * res * res + (res + 1) ** 2)
    print(f'blur_act_sform {C:3d} ch @{res:4d}: {ms * 1e3:8.1f} us  {byts / ms / 1e6:7.1f} GB/s', flush=True)
    del z, nz, ys

# third block: the plain-conv layers' activation backward -> S-form (`act_bwd_sform`, with the ToRGB branch): in the W+ loop
# only the 1024² layer and the low resolutions run it (the others are fused into the stride-2 conv's epilogue)
for res in (64, 256, 1024):
    C = CH[res]
    g = torch.Generator().manual_seed(res)
    out = torch.randn(B, C, res, res, generator=g).to(dev)
    nz = torch.randn(B, 1, res, res, generator=g).to(dev)
    nw, bias = torch.tensor([0.1], device=dev), torch.zeros(C, device=dev)
    d = (1 + 0.3 * torch.randn(B, C, generator=g)).abs().to(dev)
    mul2 = torch.tensor([2.0 ** -9, 2.0 ** 9], device=dev)
    grgb = (1e-3 * torch.randn(B, 3, res, res, generator=g)).to(dev)
    wrgb, srgb = torch.randn(3, C, generator=g).to(dev), (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    dst = ops.SForm(B, C, res, res, dev)
    ms = timeit(lambda: ops.act_bwd_producer(out, None, nz, nw, bias, d, mul2, dst, g_rgb=grgb, w_rgb=wrgb, s_rgb=srgb))
    byts = 4.0 * B * C * 2 * res * res
    print(f'act_bwd_sform {C:3d} ch @{res:4d}: {ms * 1e3:8.1f} us  {byts / ms / 1e6:7.1f} GB/s', flush=True)
    del out, nz, dst

# fourth block: ToRGB that also writes the S-form input of the following up-sampling conv (`torgb_fwd_sform`, 128² .. 512²)
k4 = (k1[:, None] * k1[None, :] / 64 * 4).contiguous().to(dev)
for res in (128, 256, 512):
    C = CH[res]
    g = torch.Generator().manual_seed(res)
    x = torch.randn(B, C, res, res, generator=g).to(dev)
    w = torch.randn(3, C, generator=g).to(dev)
    s, s2 = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev), (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
    skip = torch.randn(B, 3, res // 2, res // 2, generator=g).to(dev)
    bias = torch.zeros(3, device=dev)
    ys = ops.SForm(B, C, res, res, dev)
    ms = timeit(lambda: ops.torgb(x, w, s, bias, skip, k4, ys=ys, ys_scale=s2))
    byts = 4.0 * B * C * 2 * res * res
    print(f'torgb_fwd_sform {C:3d} ch @{res:4d}: {ms * 1e3:8.1f} us  {byts / ms / 1e6:7.1f} GB/s', flush=True)
    del x, ys


# the F-form tail of the 1024² level: tile kernel against the strip walk
res = 1024
C, H = CH[res], res // 2
g = torch.Generator().manual_seed(res)
pitch = (res + 1 + 3) // 4 * 4
z = torch.randn(B, C, res + 1, pitch, generator=g).to(dev)
nz = torch.randn(B, 1, res, res, generator=g).to(dev)
nw, bias = torch.tensor([0.1], device=dev), torch.zeros(C, device=dev)
s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
vm = torch.zeros(B, ops.VMAX_SLOTS, dtype=torch.int32, device=dev)
for r1 in (False, True):
    ms = timeit(lambda: ops.blur_act_fform(z, kf, H, H, bias, nz, nw, act=True, ys_scale=s, vmax=vm, rank_one=r1))
    byts = 4.0 * B * C * (res * res + (res + 1) ** 2)
    print(f'blur_act_fform {C:3d} ch @{res:4d} rank_one={r1}: {ms * 1e3:8.1f} us  {byts / ms / 1e6:7.1f} GB/s', flush=True)
