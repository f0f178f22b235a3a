#!/usr/bin/env python3
"""In-kernel clock of the dominant conv (MI355X_MICROARCH.md, DVFS item 6), from the DIAGNOSTIC library
(`make -C ood-gan-inversion_amd STAMP=1` -> liboodgan_hip_stamp.so: s_memtime / s_memrealtime around the K loop of
conv_f16s_s1big_kernel; the production library has no stamps):

    OODGAN_LIB=ood-gan-inversion_amd/oodgan/liboodgan_hip_stamp.so python tools/clock_probe.py

  (a) back to back: the 512 -> 512 @64² forward instance (B = 8, random data) launched for >= 2.5 s;
  (b) in the loop:  30 W+ steps of the bench workload (B = 8, 1024²) — the stamps of the last launches of the kernel.
clock = d(shader cycles) / d(100 MHz ticks) x 100 MHz per workgroup; median / p10 / p90 over the workgroups."""
import ctypes
import json
import math
import os
import sys
import time

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import _lib, ops, synth  # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
assert hasattr(L, 'oodgan_debug_set_stamp_buffer'), 'load the stamp build: OODGAN_LIB=.../liboodgan_hip_stamp.so'
NS = 1 << 16
stamps = torch.zeros(NS, 2, dtype=torch.int64, device=dev)
L.oodgan_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_long]
assert L.oodgan_debug_set_stamp_buffer(stamps.data_ptr(), NS) == 0


def clocks(n_wg):
    s = stamps[:n_wg].cpu().double()
    ok = s[:, 1] > 0
    ghz = (s[ok, 0] / s[ok, 1]) * 0.1
    us = s[ok, 1] / 100.0
    q = lambda v, p: float(torch.quantile(v, p))
    return dict(workgroups=int(ok.sum()), clock_GHz_median=round(q(ghz, 0.5), 3), clock_GHz_p10=round(q(ghz, 0.1), 3),
                clock_GHz_p90=round(q(ghz, 0.9), 3), kloop_us_median=round(q(us, 0.5), 2))


out = {}
g = torch.Generator().manual_seed(1)
B, C, H = 8, 512, 64
w = (torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9)).to(dev)
x = torch.randn(B, C, H, H, generator=g).to(dev)
s = (1 + 0.3 * torch.randn(B, C, generator=g)).to(dev)
xs = ops.to_sform(x, s)
wf = ops.pack_conv3x3(w, precision='f16s')
conv = lambda: ops.conv3x3(xs, wf, C, ops.CONV_S1, out_scale=s)
conv()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 2.5:
    for _ in range(50):
        conv()
    torch.cuda.synchronize()
    n += 50
dt = time.perf_counter() - t0
stamps.zero_()
for _ in range(20):
    conv()
torch.cuda.synchronize()
out['back_to_back'] = dict(clocks((H // 16) * (H // 32) * B * (C // 64)), launches=n, us_per_launch=round(dt / n * 1e6, 1),
                           layer='512->512 @64x64 forward, B=8')

from oodgan.engine import GeneratorEngine, WPlusInverter  # noqa: E402
size = 1024
P = synth.generator_state(size, seed=0)
eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(17)]
w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
inv = WPlusInverter(eng)
inv.invert(target, w0, noises, steps=3)
stamps.zero_()
torch.cuda.synchronize()
t0 = time.perf_counter()
inv.invert(target, w0, noises, steps=30)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# the last launch of the kernel in a step is the input gradient of the 64² layer (256 workgroups); larger grids of earlier
# launches left their stamps in the slots above
out['in_loop_last_64'] = dict(clocks(256), ms_per_step=round(dt / 30 * 1e3, 2), note='slots 0-255: the 64x64 input-gradient instance (last s1big launch of a step)')
out['in_loop_slots_256_4095'] = dict(clocks(4096), note='slots 0-4095: mixture of the 64² ... 512² instances')
print(json.dumps(out))
