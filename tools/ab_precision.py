#!/usr/bin/env python3
"""Same-process A/B of the W+ loop (bench workload: B=8, 1024², bench.py's synthetic inputs) between conv arithmetics:
`python tools/ab_precision.py [steps] [streams] [precisions]` runs the loop alternately per precision (three rounds each, the first
is a warm-up) and prints ms per W+ step and the loss curve's end — 'f16s' (three matrix instructions per product everywhere) against
'f16s-g2' (two in the input-gradient convs, include/oodgan.h x_hi_only)."""
import os
import sys
import time

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
from oodgan import synth  # noqa: E402
from oodgan.engine import GeneratorEngine, WPlusInverter  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 1
precs = sys.argv[3].split(',') if len(sys.argv) > 3 else ['f16s', 'f16s-g2']
B, size = 8, 1024
dev = torch.device('cuda:0')
state = {k: v.to(dev) for k, v in synth.generator_state(size, seed=0).items()}
engs = {p: GeneratorEngine(state, size, precision=p) for p in precs}
target = torch.cat([synth.make_images(size, 1, seed=1000 + i) for i in range(B)]).to(dev)
noises = [torch.cat([synth.make_noises(size, 1, seed=2000 + i)[k] for i in range(B)]).to(dev) for k in range(17)]
w0 = torch.cat([synth.make_latents(size, 1, seed=3000 + i, std=0.3) for i in range(B)]).to(dev)
res = {p: [] for p in precs}
for rnd in range(3):
    for p in precs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w, l = WPlusInverter(engs[p]).invert(target, w0, noises, steps=steps, streams=streams)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        res[p].append(ms)
        print(f'round {rnd} {p:8s}: {ms:7.3f} ms per W+ step ({steps} steps, {streams} streams), loss[0] {l[0].mean().item():.6f} loss[-1] {l[-1].mean().item():.6f}', flush=True)
for p in precs:
    print(f'{p:8s}: best {min(res[p][1:]):.3f} ms per step')
