#!/usr/bin/env python3
"""Timing of the one-pass up-conv of the 1024² level (csrc/conv_f16s_upvb.hip) at the production shape (B=8, 64 -> 32 channels, 512² -> 1024²)
next to the two passes it replaces (transposed conv + blur_act_fform); OODGAN_LIB selects another build (ablation variants)."""
import math
import os
import sys

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import ops  # noqa: E402

dev = torch.device('cuda:0')
B, K, M, H = 8, 64, 32, 512
g = torch.Generator().manual_seed(1)
x = torch.randn(B, K, H, H, generator=g).to(dev)
s = (1 + 0.3 * torch.randn(B, K, generator=g)).to(dev)
d = (1 + 0.3 * torch.randn(B, M, generator=g)).abs().to(dev)
w = torch.randn(M, K, 3, 3, generator=g).to(dev)
k1 = torch.tensor([1., 3., 3., 1.])
k = (k1[:, None] * k1[None, :] / 64 * 4).contiguous().to(dev)
nz = torch.randn(B, 1, 2 * H, 2 * H, generator=g).to(dev)
nw, bias = torch.tensor([0.1], device=dev), torch.zeros(M, device=dev)
xs = ops.to_sform(x, s)
del x
wpk = ops.pack_conv3x3(w, 1 / math.sqrt(K * 9), precision='f16s')
wvb = ops.pack_upconv_vblur(w, 1 / math.sqrt(K * 9), k)
vm = torch.zeros(B, ops.VMAX_SLOTS, dtype=torch.int32, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


from oodgan import _lib  # noqa: E402
for nwv in (12, 6, 4):
    _lib.set_tunable('upvb_waves', nwv)
    t1 = timeit(lambda: ops.upconv_vblur_fform(xs, wvb, out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=True, ys_scale=d, vmax=vm))
    print(f'lib {os.environ.get("OODGAN_LIB", "product")}: one pass, upvb_waves {nwv}: {t1:.1f} us', flush=True)
if not os.environ.get('OODGAN_LIB'):
    def two():
        z = ops.conv3x3(xs, wpk, M, ops.CONV_T2, out_scale=d)
        return ops.blur_act_fform(z, k, H, H, bias, nz, nw, act=True, ys_scale=d, vmax=vm, rank_one=True)
    print(f'two passes {timeit(two):.1f} us  (algorithmic bytes: x {B * K * H * H * 4 / 1e9:.2f} GB + y {B * M * 4 * H * H * 4 / 1e9:.2f} GB)')
