#!/usr/bin/env python3
"""Same-process A/B of an engine switch on the bench workload (B=8, 1024², one stream): alternates the W+ loop with the switch on and
off, three rounds.  Usage: python tools/ab_engine_flag.py fform [steps]   (switches: an engine attribute such as fuse_x; fform = F-form hand-off of the last styled conv; tiny = ops.USE_TINY)"""
import os
import sys
import time

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import ops, synth  # noqa: E402
from oodgan.engine import GeneratorEngine, WPlusInverter  # noqa: E402

flag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda:0')
size, B = 1024, 8
P = synth.generator_state(size, seed=0)
eng = GeneratorEngine({k: v.to(dev) for k, v in P.items()}, size)
target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(17)]
w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
inv = WPlusInverter(eng)


def set_flag(on):
    if flag == 'tiny':
        ops.USE_TINY = on
    elif flag == 'fform':
        eng.layers_styled_last = [L for L in eng.layers if L.kind != 'rgb'][-1] if on else None
    else:
        setattr(eng, flag, on)


inv.invert(target, w0, noises, steps=3)
for rnd in range(3):
    for on in (True, False):
        set_flag(on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w, l = inv.invert(target, w0, noises, steps=steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'{flag}={"on " if on else "off"} round {rnd + 1}: {dt / steps * 1e3:.3f} ms per W+ step, final loss {l[-1].mean().item():.6f}', flush=True)
