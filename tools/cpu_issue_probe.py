#!/usr/bin/env python3
"""How long does the HOST need to issue one W+ step (≈170 launches through ctypes)?  The GPU is parked behind a sleep kernel, ten steps
are issued behind it, and the time at which the issue loop ends (the first host read-back) is taken.  With three streams the host issues
three sub-batch steps per step of the job: if 3 x this figure approaches the GPU time of a step, the streams are host-bound.
`python tools/cpu_issue_probe.py [B]`"""
import os
import sys
import time

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'ood-gan-inversion_amd'))
import torch  # noqa: E402
from oodgan import synth  # noqa: E402
from oodgan.engine import GeneratorEngine, WPlusInverter  # noqa: E402

dev = torch.device('cuda:0')
size, B = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = GeneratorEngine({k: v.to(dev) for k, v in synth.generator_state(size, seed=5).items()}, size)
target = synth.make_images(size, B, seed=9).to(dev)
noises = [n.to(dev) for n in synth.make_noises(size, B, seed=7)]
w0 = synth.make_latents(size, B, seed=14).to(dev)
inv = WPlusInverter(eng)
inv.invert(target, w0, noises, steps=5)
torch.cuda.synchronize()
mark = {}
orig = eng.bwd_scale_violated
def hooked():
    mark.setdefault('t', time.perf_counter())
    return orig()
eng.bwd_scale_violated = hooked
N = 10
torch.cuda._sleep(int(6e9))          # ~3 s: the GPU does not start before the host is done issuing
t0 = time.perf_counter()
inv.invert(target, w0, noises, steps=N)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'B={B}: host issue time {1e3 * (mark["t"] - t0) / N:.2f} ms per W+ step ({N} steps issued behind a parked GPU)')
