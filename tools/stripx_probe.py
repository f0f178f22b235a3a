#!/usr/bin/env python3
"""Where a wave of the 1024² strip convs (csrc/conv_f16s_stripx.hip) spends its cycles, from the DIAGNOSTIC library
(`make -C ood-gan-inversion_amd STAMP=1` -> liboodgan_hip_stamp.so: s_memtime around the phases of the tile loop, summed per wave):

    OODGAN_LIB=ood-gan-inversion_amd/oodgan/liboodgan_hip_stamp.so python tools/stripx_probe.py

B = 8, 32 -> 32 channels at 1024²: forward (x_fform 1, fused ToRGB) and input gradient (x_fform 2).  Per tile and wave, medians over
the 1024 waves: counted vmcnt wait, barrier, matrix phase with the woven conversion / epilogue arithmetic, stores; clock."""
import ctypes
import math
import os
import sys

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
assert hasattr(L, 'oodgan_debug_set_stripx_stamp_buffer'), 'load the stamp build: OODGAN_LIB=.../liboodgan_hip_stamp.so'
NWG = 256
st = torch.zeros(NWG, 4, 6, dtype=torch.int64, device=dev)
L.oodgan_debug_set_stripx_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert L.oodgan_debug_set_stripx_stamp_buffer(st.data_ptr(), NWG) == 0


def to_fform(x):
    B_, C_, H_, W_ = x.shape
    return ops.FForm(x.view(B_, C_ // 16, 16, H_, W_).permute(0, 1, 3, 4, 2).contiguous().view(B_, C_, H_, W_))


B, C, H = 8, 32, 1024
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
xf = to_fform(x)
o2f = to_fform(torch.randn(B, C, H, H, generator=g).to(dev))


def report(tag, fn):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    v = st.cpu().double().reshape(-1, 6)
    tiles = H // 4 + 1
    ghz = v[:, 4] / v[:, 5] * 0.1
    med = lambda t: float(torch.median(t))
    us = lambda cyc: med(cyc / ghz) / 1e3 / tiles
    print(f'{tag}: per tile and wave, us (median of {v.shape[0]} waves): wait {us(v[:, 0]):.3f}  barrier {us(v[:, 1]):.3f}  matrix phase {us(v[:, 2]):.3f}  '
          f'stores {us(v[:, 3]):.3f}  | loop {med(v[:, 5]) / 100:.1f} us at {med(ghz):.2f} GHz', flush=True)


report('forward (x_fform 1, rgb)', lambda: ops.conv3x3(xf, wf, C, ops.CONV_S1, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_weight=nw, act=ops.ACT_LRELU, rgb=(w_rgb, s_rgb)))
report('input gradient (x_fform 2)', lambda: ops.conv3x3(o2f, wb, C, ops.CONV_S1, out_scale=s, dotx=xf, in_mul2=mul2, dot_actgrad=ops.DotActGrad(),
                                                          xf_act=ops.ActBwdX(nz, nw, bias, d, mul2, g_rgb, w_rgb, s_rgb)))
