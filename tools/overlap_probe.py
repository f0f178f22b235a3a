#!/usr/bin/env python3
"""Do an MFMA-bound conv (8-wave stride-1 kernel, 155 KB of LDS per workgroup: one per CU) and an HBM-bound kernel share the
GPU when they are launched on two HIP streams?  Times N launches of each alone and both together.  The HBM-bound side is
`to_sform` (no LDS), `act_bwd_blurT` PRE strip walk (19 KB of LDS) or `blur_act_sform` (tile kernel)."""
import math
import os
import sys
import time

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import ops  # noqa: E402

dev = torch.device('cuda:0')
B, N = 8, 20
g = torch.Generator().manual_seed(1)
C, H = 256, 128
w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
x = torch.randn(B, C, H, H, generator=g).to(dev)
s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
xs = ops.to_sform(x, s)
wf = ops.pack_conv3x3(w, precision='f16s')
conv = lambda: ops.conv3x3(xs, wf, C, ops.CONV_S1, out_scale=s)

res, Cm = 1024, 32
big = torch.randn(B, Cm, res, res, generator=g).to(dev)
sm = (1 + 0.3 * torch.randn(B, Cm, generator=g)).to(dev)
dst = ops.SForm(B, Cm, res, res, dev)
mem_a = lambda: ops.to_sform(big, sm, out=dst)
k1 = torch.tensor([1., 3., 3., 1.])
kfl = (k1[:, None] * k1[None, :] / 64 * 4).flip(0, 1).contiguous().to(dev)
d = (1 + 0.3 * torch.randn(B, Cm, generator=g)).abs().to(dev)
mul2 = torch.tensor([2.0 ** -9, 2.0 ** 9], device=dev)
gf = (1e-3 * torch.randn(B, Cm, res, res, generator=g)).to(dev)
nz = torch.randn(B, 1, res, res, generator=g).to(dev)
nw, bias = torch.tensor([0.1], device=dev), torch.zeros(Cm, device=dev)
ph = ops.SFormPhases(B, Cm, res // 2, res // 2, dev)
link = ops.DotActGrad()
link.dot_part = torch.zeros(B, Cm, 4, device=dev)
link.scale = sm
mem_b = lambda: ops.act_bwd_producer(None, gf, nz, nw, bias, d, mul2, ph, blur_kernel=kfl, dot_of=link)

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(fa, fb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa:
        with torch.cuda.stream(sa):
            for _ in range(N):
                fa()
    if fb:
        with torch.cuda.stream(sb):
            for _ in range(N):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / N


for name, fb in (('to_sform (no LDS)', mem_a), ('act_bwd_blurT PRE strip (19 KB LDS)', mem_b)):
    for _ in range(2):
        run(conv, fb)
    ta, tb, tab = run(conv, None), run(None, fb), run(conv, fb)
    print(f'{name}: conv alone {ta:.3f} ms, producer alone {tb:.3f} ms, both on two streams {tab:.3f} ms per pair '
          f'(serial {ta + tb:.3f}, perfect overlap {max(ta, tb):.3f})', flush=True)
