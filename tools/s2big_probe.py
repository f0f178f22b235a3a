#!/usr/bin/env python3
"""Where a workgroup of the fused stride-2 conv (conv_f16s_s2big_kernel<true, MH, true>) spends its time, from the DIAGNOSTIC library
(`make -C ood-gan-inversion_amd STAMP=1`): s_memtime stamps at the phase boundaries of waves 0 and 7 of every workgroup.
    OODGAN_LIB=ood-gan-inversion_amd/oodgan/liboodgan_hip_stamp.so python tools/s2big_probe.py [res ...]       (default 1024 512)"""
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
assert hasattr(L, 'oodgan_debug_set_s2big_stamp_buffer'), 'load the stamp build: OODGAN_LIB=.../liboodgan_hip_stamp.so'
L.oodgan_debug_set_s2big_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_long]
NS = 1 << 14
CH = {64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
B = 8
for res in [int(v) for v in sys.argv[1:]] or [1024, 512]:
    cin, cout = CH[res // 2], CH[res]
    H = res // 2
    g = torch.Generator().manual_seed(res)
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
    nz = torch.randn(B, 1, H, H, generator=g).to(dev)
    grgb = (1e-3 * torch.randn(B, 3, H, H, generator=g)).to(dev)
    wrgb, srgb = torch.randn(3, cin, generator=g).to(dev), (1 + 0.3 * torch.randn(B, cin, generator=g)).to(dev)
    dl = (1 + 0.3 * torch.randn(B, cin, generator=g)).abs().to(dev)
    mul2 = torch.tensor([2.0 ** -9, 2.0 ** 9], device=dev)
    dst = ops.SForm(B, cin, H, H, dev)
    xsf = ops.SFormSaved(ops.to_sform(x, s), s)          # the saved activation as its consumer's S-form (the loop's form)

    def fused(dotx):
        fz = ops.ActBwdFusion(dst, nz, torch.tensor([0.1], device=dev), torch.zeros(cin, device=dev), dl, mul2, g_rgb=grgb, w_rgb=wrgb, s_rgb=srgb)
        ops.conv3x3(gp, wb, cin, ops.CONV_S2, out_scale=s, dotx=dotx, fuse=fz, want_y=False)

    for tag, dotx in (('dotx fp32', x), ('dotx S-form', xsf)):
        stamps = torch.zeros(NS, 2, 8, dtype=torch.int64, device=dev)
        assert L.oodgan_debug_set_s2big_stamp_buffer(stamps.data_ptr(), NS) == 0
        for _ in range(3):
            fused(dotx)
        torch.cuda.synchronize()
        st = stamps.cpu().double()
        ok = st[:, 0, 6] > 0
        n = int(ok.sum())
        names = ['setup + early loads', 'K loop', 'wait for / decode the saved activations', 'M-tile 0 (arithmetic, sums, stores)', 'M-tile 1', 'final reduction']
        print(f'{cout}->{cin} @{res}->{H}, {tag}: {n} workgroups stamped')
        for wv in (0, 1):
            seg = [(st[ok, wv, i + 1] - st[ok, wv, i]) for i in range(6)]
            tot = st[ok, wv, 6] - st[ok, wv, 0]
            line = ', '.join(f'{nm} {float(torch.median(v)):.0f}' for nm, v in zip(names, seg))
            print(f'  wave {0 if wv == 0 else 7}: total {float(torch.median(tot)):.0f} cycles (p90 {float(torch.quantile(tot, 0.9)):.0f}); {line}')
